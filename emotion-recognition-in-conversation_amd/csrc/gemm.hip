// K2: dense GEMMs of the path on the gfx950 matrix cores.
//
//  * gemm_f32_kernel  : v_mfma_f32_16x16x4_f32 (exact fp32 fma chains) -- the
//    parity path for every nn.Linear / matmul and its two gradients.
//  * gemm_bf16x_kernel: v_mfma_f32_16x16x32_bf16 with the big streamed operand
//    (padded feature block) stored in bf16, the small operand converted while
//    staging; fp32 accumulate.
//
// Geometry (both): 256 threads = 4 wavefronts stacked along M; workgroup tile
// 64 x (16*NF); K consumed in chunks of BK staged through LDS with a register
// prefetch of the next chunk.  NF = 2 keeps the per-wave tile small so that the
// skinny matrices of this workload (M ~ 2k nodes, N ~ 100) still give > 1000
// wavefronts; K is additionally split over blockIdx.z into partial slabs that
// the caller reduces (deterministic, no float atomics).
//
// Fragment maps (cdna_hip_programming.md section 3): 16x16x4 f32: lane l holds
// A[l&15][l>>4], B[l>>4][l&15]; C/D reg i of lane l is C[4*(l>>4)+i][l&15].
#include "erc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

struct GemmP {
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
    int chunks_per_split;
    int ones_col, act, accumulate;
    int a_vec, b_vec;
    float act_scale, drop_p;
    // grouped form 0 only: cross-modal entries of MMGCN's adjacency added in the epilogue (xr_CR != null)
    const float* xr_CR;   // [B][M*M][P]
    const float* xr_h;    // the un-offset B operand (node rows, pitch ldb)
    int xr_M, xr_N, xr_P, xr_m, xr_b, xr_off;
    // planes > 1: the K axis runs over `planes` operand planes of K columns each (A plane p at A + p * a_plane, B at
    // B + p * b_plane): sum_p A_p B_p^T in ONE launch -- the per-layer products of MMGCN's backward that only meet in a sum
    int planes;
    int64_t a_plane, b_plane;
};

constexpr int BM = 64;
constexpr int BK = 32;
constexpr int KC_STRIDE = BK + 2;  // K-contiguous LDS rows: bank = 2*row + k, conflict-free fragment reads

__device__ __forceinline__ float4 ld4_guard(const float* p, int valid, bool vec) {
    // valid = number of in-range elements starting at p (<=0: none)
    if (valid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
    return v;
}

template <int A_KMAJOR, int B_KMAJOR, int NF>
__device__ __forceinline__ void gemm_f32_body(const GemmP& p, const int bx, const int by, const int z) {
    constexpr int BN = 16 * NF;
    constexpr int A_ELEMS = A_KMAJOR ? BK * (BM + 16) : BM * KC_STRIDE;
    constexpr int B_ELEMS = B_KMAJOR ? BK * (BN + 16) : BN * KC_STRIDE;
    constexpr int B_PASS = (BN * BK) / (256 * 4);  // float4 per thread for the B tile
    static_assert(B_PASS >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) float lds[A_ELEMS + B_ELEMS];
    float* As = lds;
    float* Bs = lds + A_ELEMS;

    const float* __restrict__ A = (const float*)p.A;
    const float* __restrict__ B = (const float*)p.B;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = bx * BN, m0 = by * BM;
    const int cpl = (p.K + BK - 1) / BK;                     // chunks per plane
    const int nchunk = cpl * (p.planes > 1 ? p.planes : 1);
    const int c_begin = z * p.chunks_per_split;
    const int c_end = min(nchunk, c_begin + p.chunks_per_split);
    const float* const A0 = A;
    const float* const B0 = B;

    float4 ra[2], rb[B_PASS];

    auto load_chunk = [&](int c) {
        int k0 = c * BK;
        const float* A = A0;
        const float* B = B0;
        if (p.planes > 1) {
            const int pl = c / cpl;
            k0 = (c - pl * cpl) * BK;
            A = A0 + (int64_t)pl * p.a_plane;
            B = B0 + (int64_t)pl * p.b_plane;
        }
        if (!A_KMAJOR) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (tid >> 3) + 32 * i, kk = k0 + 4 * (tid & 7);
                const int m = m0 + row;
                if (m < p.M) {
                    const int64_t src = p.a_gather ? (int64_t)p.a_gather[m] : (int64_t)m;
                    ra[i] = ld4_guard(A + src * p.lda + kk, p.K - kk, p.a_vec);
                } else if (m == p.M && p.ones_col == 2) {
                    const int v = p.K - kk;
                    ra[i] = make_float4(v > 0 ? 1.f : 0.f, v > 1 ? 1.f : 0.f, v > 2 ? 1.f : 0.f, v > 3 ? 1.f : 0.f);
                } else
                    ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kr = (tid >> 4) + 16 * i, mm = m0 + 4 * (tid & 15);
                const int k = k0 + kr;
                if (k < p.K) {
                    float4 v = ld4_guard(A + (int64_t)k * p.lda + mm, p.M - mm, p.a_vec);
                    if (p.ones_col == 2) {
                        if (mm == p.M) v.x = 1.f;
                        if (mm + 1 == p.M) v.y = 1.f;
                        if (mm + 2 == p.M) v.z = 1.f;
                        if (mm + 3 == p.M) v.w = 1.f;
                    }
                    ra[i] = v;
                } else
                    ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (!B_KMAJOR) {
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) {
                const int row = (tid >> 3) + 32 * i, kk = k0 + 4 * (tid & 7);
                const int n = n0 + row;
                if (n < p.N)
                    rb[i] = ld4_guard(B + (int64_t)n * p.ldb + kk, p.K - kk, p.b_vec);
                else if (n == p.N && p.ones_col == 1) {
                    const int v = p.K - kk;
                    rb[i] = make_float4(v > 0 ? 1.f : 0.f, v > 1 ? 1.f : 0.f, v > 2 ? 1.f : 0.f, v > 3 ? 1.f : 0.f);
                } else
                    rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            constexpr int TPR = BN / 4;        // threads per k-row
            constexpr int RPP = 256 / TPR;     // k-rows per pass
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) {
                const int kr = tid / TPR + RPP * i, nn = n0 + 4 * (tid % TPR);
                const int k = k0 + kr;
                if (k < p.K) {
                    const int64_t src = p.b_gather ? (int64_t)p.b_gather[k] : (int64_t)k;
                    float4 v = ld4_guard(B + src * p.ldb + nn, p.N - nn, p.b_vec);
                    if (p.ones_col == 1) {
                        if (nn == p.N) v.x = 1.f;
                        if (nn + 1 == p.N) v.y = 1.f;
                        if (nn + 2 == p.N) v.z = 1.f;
                        if (nn + 3 == p.N) v.w = 1.f;
                    }
                    rb[i] = v;
                } else
                    rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    auto store_chunk = [&]() {
        if (!A_KMAJOR) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float* d = As + ((tid >> 3) + 32 * i) * KC_STRIDE + 4 * (tid & 7);
                *reinterpret_cast<float2*>(d) = make_float2(ra[i].x, ra[i].y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(ra[i].z, ra[i].w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<float4*>(As + ((tid >> 4) + 16 * i) * (BM + 16) + 4 * (tid & 15)) = ra[i];
        }
        if (!B_KMAJOR) {
#pragma unroll
            for (int i = 0; i < B_PASS; ++i) {
                float* d = Bs + ((tid >> 3) + 32 * i) * KC_STRIDE + 4 * (tid & 7);
                *reinterpret_cast<float2*>(d) = make_float2(rb[i].x, rb[i].y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(rb[i].z, rb[i].w);
            }
        } else {
            constexpr int TPR = BN / 4;
            constexpr int RPP = 256 / TPR;
#pragma unroll
            for (int i = 0; i < B_PASS; ++i)
                *reinterpret_cast<float4*>(Bs + (tid / TPR + RPP * i) * (BN + 16) + 4 * (tid % TPR)) = rb[i];
        }
    };

    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (c_begin < c_end) load_chunk(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        store_chunk();
        __syncthreads();
        if (c + 1 < c_end) load_chunk(c + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int k = 4 * ks + g;
            const float a = A_KMAJOR ? As[k * (BM + 16) + 16 * w + r] : As[(16 * w + r) * KC_STRIDE + k];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float b = B_KMAJOR ? Bs[k * (BN + 16) + 16 * f + r] : Bs[(16 * f + r) * KC_STRIDE + k];
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[f], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue
    float* __restrict__ C = p.C + (int64_t)z * p.c_slab;
    uint64_t rng_off = 0, rng_seed = 0;
    if (p.act == 3) {
        rng_off = p.rng[0];
        rng_seed = p.rng[1];
    }
    // MMGCN cross-modal terms: out[(m,t), :] += sum_{n != m} CR[b][m*M+n][t] * h[(n,t), :]  (the off-block-diagonal
    // entries of create_big_adj, mmgcn_models.py:617-640).  All operands of this lane's elements are requested up front.
    if (p.xr_CR) {   // uniform; two or three modalities: the (at most two) other ones, in ascending order
        const int M_ = p.xr_M, m_ = p.xr_m;
        const int na = m_ == 0 ? 1 : 0, nb = (M_ > 2) ? (m_ == 2 ? 1 : 2) : na;
        const float wb = M_ > 2 ? 1.f : 0.f;
        float ca[4], cb[4], ha[NF][4], hb[NF][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = min(m0 + 16 * w + 4 * g + i, p.M - 1);
            const int64_t cbase = ((int64_t)p.xr_b * M_ * M_ + m_ * M_) * p.xr_P + t;
            ca[i] = p.xr_CR[cbase + (int64_t)na * p.xr_P];
            cb[i] = p.xr_CR[cbase + (int64_t)nb * p.xr_P];
            const float* hra = p.xr_h + ((int64_t)na * p.xr_N + p.xr_off + t) * p.ldb;
            const float* hrb = p.xr_h + ((int64_t)nb * p.xr_N + p.xr_off + t) * p.ldb;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int cc = min(n0 + 16 * f + r, p.N - 1);
                ha[f][i] = hra[cc], hb[f][i] = hrb[cc];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f][i] = (acc[f][i] + ca[i] * ha[f][i]) + wb * cb[i] * hb[f][i];
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int col = n0 + 16 * f + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + 16 * w + 4 * g + i;
            float v = acc[f][i];
            if (row >= p.M) {
                if (row == p.M && p.ones_col == 2 && col < p.N && p.bias_out)
                    p.bias_out[(int64_t)z * p.bias_slab + col] = v;
                continue;
            }
            if (col < p.N) {
                float* dst = C + (int64_t)row * p.ldc + col;
                if (p.bias) v += p.bias[col];
                if (p.accumulate) v += *dst;  // accumulate BEFORE the activation: relu(C_old + A B + bias)
                if (p.act == 1)
                    v = fmaxf(v, 0.f);
                else if (p.act == 2)
                    v = p.aux[(int64_t)row * p.ldaux + col] > 0.f ? v * p.act_scale : 0.f;
                else if (p.act == 3) {
                    const float u = erc_uniform(rng_seed, rng_off, (uint64_t)row * (uint64_t)p.N + col);
                    v = (u >= p.drop_p) ? fmaxf(v, 0.f) * p.act_scale : 0.f;
                }
                *dst = v;
            } else if (col == p.N && p.ones_col == 1 && p.bias_out) {
                p.bias_out[(int64_t)z * p.bias_slab + row] = v;
            }
        }
    }
}

template <int A_KMAJOR, int B_KMAJOR, int NF>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p) {
    gemm_f32_body<A_KMAJOR, B_KMAJOR, NF>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped products over the per-(dialogue, modality) blocks of MMGCN's adjacency (mmgcn_models.py:582-646):
// group z = b*n_mod + m covers the L_b nodes [m*n_nodes + node_off[b], +L_b) and the block blk + z*pitch*pitch.
//   FORM 0 ("block x nodes"):  C_nodes[L,N] (+)= Blk[L,L] * B_nodes[L,N]
//   FORM 1 ("nodes x nodes^T"): Blk[L,L]    (+)= A_nodes[L,K] * B_nodes[L,K]^T
struct GroupP {
    const int32_t* node_off;
    int n_mod, n_nodes, pitch;
    int split;      // form 1: the chunk range is cut into `split` parts, part s writes slab s (C + s * c_slab)
};

template <int FORM>
__global__ __launch_bounds__(256) void gemm_f32_grouped_kernel(GemmP p, GroupP g) {
    const int z = (int)blockIdx.z / g.split, sp = (int)blockIdx.z % g.split, b = z / g.n_mod, m = z % g.n_mod;
    const int off = g.node_off[b], L = g.node_off[b + 1] - off;
    if ((int)blockIdx.y * BM >= L) return;
    const int64_t row0 = (int64_t)m * g.n_nodes + off;
    const int64_t blk = (int64_t)z * g.pitch * g.pitch;
    if (FORM == 0) {
        if ((int)blockIdx.x * 32 >= p.N) return;
        p.A = (const float*)p.A + blk;
        p.xr_h = (const float*)p.B;
        p.xr_m = m, p.xr_b = b, p.xr_off = off;
        p.B = (const float*)p.B + row0 * p.ldb;
        p.C = p.C + row0 * p.ldc;
        if (p.aux) p.aux = p.aux + row0 * p.ldaux;
        p.M = L;
        p.K = L;
        p.chunks_per_split = (L + BK - 1) / BK;
        gemm_f32_body<0, 1, 2>(p, blockIdx.x, blockIdx.y, 0);
    } else {
        if ((int)blockIdx.x * 32 >= L) return;
        p.A = (const float*)p.A + row0 * p.lda;
        p.B = (const float*)p.B + row0 * p.ldb;
        p.C = p.C + blk;
        p.M = L;
        p.N = L;
        gemm_f32_body<0, 0, 2>(p, blockIdx.x, blockIdx.y, sp);
    }
}

// ---------------------------------------------------------------------------
// bf16 variant.  One operand ("X", the padded feature block) lives in HBM as
// bf16; the other is fp32 and is rounded to bf16 (RNE) while staged.  LDS
// holds bf16; BK = 64 so that each fragment read is one 16-byte ds_read.
// 16x16x32 bf16: lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15], j<8.
// LDS images are always K-contiguous rows of 64 bf16 (+8 pad): K-major sources
// are transposed by the staging writes.
// ---------------------------------------------------------------------------
constexpr int BKH = 64;
constexpr int KH_STRIDE = BKH + 8;  // in bf16 elements: 144-byte rows keep 16-B alignment, break the 128-B period

__device__ __forceinline__ unsigned short f2bf(float f) {
    const __bf16 h = (__bf16)f;  // plain cast: v_cvt_pk_bf16_f32, RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, h);
}

template <int A_KMAJOR, int B_KMAJOR, int X_IS_A, int NF>
__global__ __launch_bounds__(256) void gemm_bf16x_kernel(GemmP p) {
    constexpr int BN = 16 * NF;
    __shared__ __attribute__((aligned(16))) unsigned short lds[(BM + BN) * KH_STRIDE];
    unsigned short* As = lds;
    unsigned short* Bs = lds + BM * KH_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, z = blockIdx.z;
    const int nchunk = (p.K + BKH - 1) / BKH;
    const int c_begin = z * p.chunks_per_split;
    const int c_end = min(nchunk, c_begin + p.chunks_per_split);

    // element (row, k) of an operand tile; generic scalar fetch, used for edges and K-major sources
    auto fetch = [&](bool is_a, int row_g, int k) -> unsigned short {
        // row_g: global m (A) or n (B); returns bf16 bits, 0 outside
        if (k >= p.K) return 0;
        if (is_a) {
            if (row_g >= p.M) return 0;
            if (!A_KMAJOR) {
                const int64_t src = p.a_gather ? (int64_t)p.a_gather[row_g] : (int64_t)row_g;
                if (X_IS_A) return ((const unsigned short*)p.A)[src * p.lda + k];
                return f2bf(((const float*)p.A)[src * p.lda + k]);
            } else {
                if (X_IS_A) return ((const unsigned short*)p.A)[(int64_t)k * p.lda + row_g];
                return f2bf(((const float*)p.A)[(int64_t)k * p.lda + row_g]);
            }
        } else {
            if (row_g > p.N || (row_g == p.N && p.ones_col != 1)) return 0;
            if (row_g == p.N) return 0x3F80;  // bf16(1.0): virtual ones column
            if (!B_KMAJOR) {
                if (!X_IS_A) return ((const unsigned short*)p.B)[(int64_t)row_g * p.ldb + k];
                return f2bf(((const float*)p.B)[(int64_t)row_g * p.ldb + k]);
            } else {
                const int64_t src = p.b_gather ? (int64_t)p.b_gather[k] : (int64_t)k;
                if (!X_IS_A) return ((const unsigned short*)p.B)[src * p.ldb + row_g];
                return f2bf(((const float*)p.B)[src * p.ldb + row_g]);
            }
        }
    };

    // stage one operand tile (ROWS x 64) into its K-contiguous LDS image
    auto stage = [&](bool is_a, int c) {
        const int k0 = c * BKH;
        const int ROWS = is_a ? BM : BN;
        const int base_row = is_a ? m0 : n0;
        unsigned short* dst = is_a ? As : Bs;
        const bool kmajor = is_a ? (bool)A_KMAJOR : (bool)B_KMAJOR;
        const bool is_x = is_a ? (bool)X_IS_A : !(bool)X_IS_A;
        const int vec = is_a ? p.a_vec : p.b_vec;
        if (!kmajor) {
            // rows are K-contiguous in memory: 8 threads x 8 elements per row.
            // vec: 2 = 16-byte loads, 1 = 8-byte loads (bf16 rows of 1380 elements are only 8-byte aligned)
            for (int row = tid >> 3; row < ROWS; row += 32) {
                const int kk = 8 * (tid & 7);
                const int rg = base_row + row;
                bf16x8 v;
                const int lim = is_a ? p.M : p.N;
                if (rg < lim && k0 + kk + 8 <= p.K && vec) {
                    const int64_t src = (is_a && p.a_gather) ? (int64_t)p.a_gather[rg] : (int64_t)rg;
                    const int ld = is_a ? p.lda : p.ldb;
                    if (is_x) {
                        const unsigned short* s = (const unsigned short*)(is_a ? p.A : p.B) + src * ld + k0 + kk;
                        if (vec == 2) {
                            v = *reinterpret_cast<const bf16x8*>(s);
                        } else {
                            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(s);
                            const bf16x4 hi = *reinterpret_cast<const bf16x4*>(s + 4);
                            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
                            v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                        }
                    } else {
                        const float* s = (const float*)(is_a ? p.A : p.B) + src * ld + k0 + kk;
                        const float4 lo = *reinterpret_cast<const float4*>(s);
                        const float4 hi = *reinterpret_cast<const float4*>(s + 4);
                        v[0] = f2bf(lo.x); v[1] = f2bf(lo.y); v[2] = f2bf(lo.z); v[3] = f2bf(lo.w);
                        v[4] = f2bf(hi.x); v[5] = f2bf(hi.y); v[6] = f2bf(hi.z); v[7] = f2bf(hi.w);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fetch(is_a, rg, k0 + kk + j);
                }
                *reinterpret_cast<bf16x8*>(dst + row * KH_STRIDE + kk) = v;
            }
        } else {
            // memory is K-major (row index contiguous): each thread reads 4 consecutive rows of one k
            // (one 8-byte / 16-byte load) and scatters them into the K-contiguous LDS image
            const int RQ = ROWS / 4;  // row-quads
            const int lim = is_a ? p.M : p.N;
            for (int idx = tid; idx < RQ * BKH; idx += 256) {
                const int kq = idx / RQ, rq = idx % RQ;
                const int k = k0 + kq, rg = base_row + 4 * rq;
                unsigned short e0, e1, e2, e3;
                if (k < p.K && rg + 4 <= lim && vec) {
                    const int ld = is_a ? p.lda : p.ldb;
                    const int64_t src = (!is_a && p.b_gather) ? (int64_t)p.b_gather[k] : (int64_t)k;
                    if (is_x) {
                        const bf16x4 q = *reinterpret_cast<const bf16x4*>(
                            (const unsigned short*)(is_a ? p.A : p.B) + src * ld + rg);
                        e0 = q[0]; e1 = q[1]; e2 = q[2]; e3 = q[3];
                    } else {
                        const float4 q = *reinterpret_cast<const float4*>((const float*)(is_a ? p.A : p.B) + src * ld + rg);
                        e0 = f2bf(q.x); e1 = f2bf(q.y); e2 = f2bf(q.z); e3 = f2bf(q.w);
                    }
                } else {
                    e0 = fetch(is_a, rg, k); e1 = fetch(is_a, rg + 1, k);
                    e2 = fetch(is_a, rg + 2, k); e3 = fetch(is_a, rg + 3, k);
                }
                dst[(4 * rq + 0) * KH_STRIDE + kq] = e0;
                dst[(4 * rq + 1) * KH_STRIDE + kq] = e1;
                dst[(4 * rq + 2) * KH_STRIDE + kq] = e2;
                dst[(4 * rq + 3) * KH_STRIDE + kq] = e3;
            }
        }
    };

    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c = c_begin; c < c_end; ++c) {
        stage(true, c);
        stage(false, c);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BKH / 32; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + (16 * w + r) * KH_STRIDE + 32 * ks + 8 * g);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (16 * f + r) * KH_STRIDE + 32 * ks + 8 * g);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[f], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    float* __restrict__ C = p.C + (int64_t)z * p.c_slab;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int col = n0 + 16 * f + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + 16 * w + 4 * g + i;
            if (row >= p.M) continue;
            if (col < p.N)
                C[(int64_t)row * p.ldc + col] = acc[f][i];
            else if (col == p.N && p.ones_col == 1 && p.bias_out)
                p.bias_out[(int64_t)z * p.bias_slab + row] = acc[f][i];
        }
    }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int S, int64_t stride,
                                                          const float* __restrict__ bias, int n_cols, int act,
                                                          float* __restrict__ out, int ld_out, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += slabs[(int64_t)s * stride + i];
        if (bias) v += bias[i % n_cols];
        if (act == 1) v = fmaxf(v, 0.f);
        const int64_t o = ld_out > 0 ? (i / n_cols) * ld_out + i % n_cols : i;
        out[o] = act == 4 ? out[o] + v : v;      // act 4: accumulate into out
    }
}

__global__ __launch_bounds__(256) void slab_reduce_batched_kernel(const float* __restrict__ ws, float* __restrict__ dst,
                                                                  const int64_t* __restrict__ jobs) {
    const int64_t* j = jobs + 5 * (int64_t)blockIdx.y;
    const int64_t src = j[0], stride = j[1], S = j[2], numel = j[3], doff = j[4];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int64_t s = 0; s < S; ++s) v += ws[src + s * stride + i];
        dst[doff + i] = v;
    }
}

template <int AK, int BKM>
int launch_f32(const GemmP& p, int nf, dim3 grid, hipStream_t st) {
    if (nf == 2)
        hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, 2>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<AK, BKM, 8>), grid, dim3(256), 0, st, p);
    return 0;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int erc_gemm_f32_stream(const float* A, int lda, int a_kmajor, const int32_t* a_gather, const float* B, int ldb,
                                   int b_kmajor, const int32_t* b_gather, float* C, int ldc, int M, int N, int K,
                                   int split_k, int64_t c_slab, int ones_col, float* bias_out, int64_t bias_slab,
                                   const float* bias, int act, const float* aux, int ldaux, float act_scale, float drop_p,
                                   const uint64_t* rng_state, int accumulate, void* stream);

extern "C" int erc_gemm_f32(const float* A, int lda, int a_kmajor, const int32_t* a_gather, const float* B, int ldb,
                            int b_kmajor, const int32_t* b_gather, float* C, int ldc, int M, int N, int K, int split_k,
                            int64_t c_slab, int ones_col, float* bias_out, int64_t bias_slab, const float* bias,
                            int act, const float* aux, int ldaux, float act_scale, float drop_p,
                            const uint64_t* rng_state, int accumulate, void* stream) {
    ERC_REQUIRE(A && B && C, "gemm_f32: null operand");
    ERC_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_f32: bad shape M=%d N=%d K=%d", M, N, K);
    // skinny outputs (everything but the basis-space products of DialogueGCN): register-streaming kernel
    if (N + (ones_col == 1 ? 1 : 0) <= 1024)
        return erc_gemm_f32_stream(A, lda, a_kmajor, a_gather, B, ldb, b_kmajor, b_gather, C, ldc, M, N, K, split_k, c_slab,
                                   ones_col, bias_out, bias_slab, bias, act, aux, ldaux, act_scale, drop_p, rng_state,
                                   accumulate, stream);
    ERC_REQUIRE(split_k >= 1, "gemm_f32: split_k must be >= 1");
    ERC_REQUIRE(!(a_kmajor && !b_kmajor), "gemm_f32: (A k-major, B k-contiguous) is not built");
    ERC_REQUIRE(split_k == 1 || (!bias && act == 0 && !accumulate), "gemm_f32: epilogue requires split_k == 1");
    ERC_REQUIRE(act >= 0 && act <= 3, "gemm_f32: act %d", act);
    ERC_REQUIRE(act != 2 || aux, "gemm_f32: act 2 needs aux");
    ERC_REQUIRE(act != 3 || rng_state, "gemm_f32: act 3 needs rng_state");
    ERC_REQUIRE(!(a_gather && a_kmajor) && !(b_gather && !b_kmajor), "gemm_f32: gather on a contiguous-K index only");
    GemmP p{};
    p.A = A; p.B = B; p.C = C; p.a_gather = a_gather; p.b_gather = b_gather;
    p.bias = bias; p.aux = aux; p.bias_out = bias_out; p.rng = rng_state;
    p.c_slab = c_slab; p.bias_slab = bias_slab;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
    p.M = M; p.N = N; p.K = K;
    const int nchunk = erc_cdiv(K, BK);
    ERC_REQUIRE(split_k <= nchunk, "gemm_f32: split_k %d exceeds the %d K-chunks (unwritten slabs)", split_k, nchunk);
    p.chunks_per_split = erc_cdiv(nchunk, split_k);
    p.ones_col = ones_col; p.act = act; p.accumulate = accumulate;
    p.a_vec = aligned16(A) && (lda % 4 == 0);
    p.b_vec = aligned16(B) && (ldb % 4 == 0);
    p.act_scale = act_scale; p.drop_p = drop_p;
    ERC_REQUIRE(ones_col >= 0 && ones_col <= 2, "gemm_f32: ones mode %d", ones_col);
    const int Nlog = N + (ones_col == 1 ? 1 : 0);
    const int Mlog = M + (ones_col == 2 ? 1 : 0);
    // small per-wave tiles unless the problem already fills the chip
    const int64_t wg_small = (int64_t)erc_cdiv(Nlog, 32) * erc_cdiv(Mlog, BM) * split_k;
    const int nf = (wg_small > 2048 && Nlog >= 128) ? 8 : 2;
    dim3 grid(erc_cdiv(Nlog, 16 * nf), erc_cdiv(Mlog, BM), split_k);
    // slabs that receive no chunk must still be defined: the host sizes split_k from nchunk, see capi.py
    hipStream_t st = (hipStream_t)stream;
    if (!a_kmajor && !b_kmajor)
        launch_f32<0, 0>(p, nf, grid, st);
    else if (!a_kmajor && b_kmajor)
        launch_f32<0, 1>(p, nf, grid, st);
    else
        launch_f32<1, 1>(p, nf, grid, st);
    ERC_LAUNCH_CHECK("gemm_f32");
    return ERC_OK;
}

extern "C" int erc_gemm_f32_grouped(int form, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                    int N_or_K, const int32_t* node_off, int n_dialogues, int n_mod, int n_nodes,
                                    int max_len, int pitch, int accumulate, int act, const float* aux, int ldaux,
                                    float act_scale, const float* cross, int planes, int64_t a_plane, int64_t b_plane,
                                    int split, int64_t c_slab, void* stream) {
    ERC_REQUIRE(A && B && C && node_off, "gemm_f32_grouped: null pointer");
    ERC_REQUIRE(planes <= 1 || form == 1, "gemm_f32_grouped: planes go with form 1");
    ERC_REQUIRE(split <= 1 || (form == 1 && !accumulate && c_slab > 0), "gemm_f32_grouped: split goes with form 1, plain stores into slabs");
    ERC_REQUIRE(form == 0 || form == 1, "gemm_f32_grouped: form %d", form);
    ERC_REQUIRE(n_dialogues > 0 && n_mod > 0 && n_nodes > 0 && max_len > 0 && pitch >= max_len && N_or_K > 0,
                "gemm_f32_grouped: bad sizes");
    ERC_REQUIRE(act == 0 || (act == 2 && aux && form == 0), "gemm_f32_grouped: act %d unsupported here", act);
    ERC_REQUIRE(!cross || form == 0, "gemm_f32_grouped: cross entries go with form 0");
    GemmP p{};
    p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.accumulate = accumulate; p.act = act; p.aux = aux; p.ldaux = ldaux; p.act_scale = act_scale;
    p.xr_CR = cross; p.xr_M = n_mod; p.xr_N = n_nodes; p.xr_P = pitch;
    GroupP g{node_off, n_mod, n_nodes, pitch, split > 1 ? split : 1};
    hipStream_t st = (hipStream_t)stream;
    if (form == 0) {
        p.N = N_or_K;
        p.lda = pitch;
        p.a_vec = aligned16(A) && (pitch % 4 == 0);
        p.b_vec = aligned16(B) && (ldb % 4 == 0);
        dim3 grid(erc_cdiv(N_or_K, 32), erc_cdiv(max_len, BM), n_dialogues * n_mod);
        hipLaunchKernelGGL(gemm_f32_grouped_kernel<0>, grid, dim3(256), 0, st, p, g);
    } else {
        p.K = N_or_K;
        p.ldc = pitch;
        p.planes = planes > 1 ? planes : 1, p.a_plane = a_plane, p.b_plane = b_plane;
        p.chunks_per_split = erc_cdiv(erc_cdiv(N_or_K, BK) * p.planes, g.split);
        p.c_slab = c_slab;
        p.a_vec = aligned16(A) && (lda % 4 == 0);
        p.b_vec = aligned16(B) && (ldb % 4 == 0);
        dim3 grid(erc_cdiv(max_len, 32), erc_cdiv(max_len, BM), n_dialogues * n_mod * g.split);
        hipLaunchKernelGGL(gemm_f32_grouped_kernel<1>, grid, dim3(256), 0, st, p, g);
    }
    ERC_LAUNCH_CHECK("gemm_f32_grouped");
    return ERC_OK;
}

extern "C" int erc_gemm_f32_planes(const float* A, int lda, int64_t a_plane, const float* B, int ldb, int64_t b_plane, float* C,
                                   int ldc, int M, int N, int K, int planes, int split_k, int64_t c_slab, int accumulate,
                                   void* stream) {
    ERC_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && planes >= 1 && split_k >= 1, "gemm_f32_planes: bad arguments");
    ERC_REQUIRE(split_k == 1 || !accumulate, "gemm_f32_planes: accumulate needs split_k == 1");
    GemmP p{};
    p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.planes = planes; p.a_plane = a_plane; p.b_plane = b_plane; p.c_slab = c_slab; p.accumulate = accumulate;
    const int nchunk = erc_cdiv(K, BK) * planes;
    ERC_REQUIRE(split_k <= nchunk, "gemm_f32_planes: split_k %d exceeds the %d K-chunks", split_k, nchunk);
    p.chunks_per_split = erc_cdiv(nchunk, split_k);
    p.a_vec = aligned16(A) && (lda % 4 == 0) && (a_plane % 4 == 0);
    p.b_vec = aligned16(B) && (ldb % 4 == 0) && (b_plane % 4 == 0);
    dim3 grid(erc_cdiv(N, 32), erc_cdiv(M, BM), split_k);
    hipLaunchKernelGGL((gemm_f32_kernel<0, 0, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("gemm_f32_planes");
    return ERC_OK;
}

extern "C" int erc_gemm_bf16x(const void* A, int lda, int a_kmajor, const int32_t* a_gather, const void* B, int ldb,
                              int b_kmajor, const int32_t* b_gather, int x_is_a, float* C, int ldc, int M, int N,
                              int K, int split_k, int64_t c_slab, int ones_col, float* bias_out, int64_t bias_slab,
                              void* stream) {
    ERC_REQUIRE(A && B && C, "gemm_bf16x: null operand");
    ERC_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_bf16x: bad shape M=%d N=%d K=%d", M, N, K);
    ERC_REQUIRE(split_k >= 1, "gemm_bf16x: split_k must be >= 1");
    GemmP p{};
    p.A = A; p.B = B; p.C = C; p.a_gather = a_gather; p.b_gather = b_gather;
    p.bias_out = bias_out; p.c_slab = c_slab; p.bias_slab = bias_slab;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    const int nchunk = erc_cdiv(K, BKH);
    ERC_REQUIRE(split_k <= nchunk, "gemm_bf16x: split_k %d exceeds the %d K-chunks (unwritten slabs)", split_k, nchunk);
    p.chunks_per_split = erc_cdiv(nchunk, split_k);
    p.ones_col = ones_col;
    // vector path: bf16 operand: 2 = 16-byte loads (ld % 8 == 0), 1 = 8-byte loads (ld % 4 == 0);
    // fp32 operand: non-zero = float4 loads (ld % 4 == 0)
    auto vec_of = [](const void* ptr, int ld, bool is_bf16) {
        if (!aligned16(ptr)) return 0;
        if (is_bf16) return ld % 8 == 0 ? 2 : (ld % 4 == 0 ? 1 : 0);
        return ld % 4 == 0 ? 2 : 0;
    };
    p.a_vec = vec_of(A, lda, x_is_a != 0);
    p.b_vec = vec_of(B, ldb, x_is_a == 0);
    ERC_REQUIRE(ones_col == 0 || ones_col == 1, "gemm_bf16x: ones mode %d", ones_col);
    const int Nlog = N + (ones_col ? 1 : 0);
    dim3 grid(erc_cdiv(Nlog, 32), erc_cdiv(M, BM), split_k);
    hipStream_t st = (hipStream_t)stream;
    if (x_is_a && !a_kmajor && !b_kmajor)
        hipLaunchKernelGGL((gemm_bf16x_kernel<0, 0, 1, 2>), grid, dim3(256), 0, st, p);
    else if (!x_is_a && a_kmajor && b_kmajor)
        hipLaunchKernelGGL((gemm_bf16x_kernel<1, 1, 0, 2>), grid, dim3(256), 0, st, p);
    else {
        erc_set_error("gemm_bf16x: only (X=A, NT) and (X=B, TN) are built");
        return ERC_E_ARG;
    }
    ERC_LAUNCH_CHECK("gemm_bf16x");
    return ERC_OK;
}

extern "C" int erc_slab_reduce(const float* slabs, int S, int64_t slab_stride, const float* bias, int n_cols, int act,
                               float* out, int ld_out, int64_t numel, void* stream) {
    ERC_REQUIRE(slabs && out && S >= 1 && numel > 0, "slab_reduce: bad arguments");
    ERC_REQUIRE((!bias && ld_out <= 0) || n_cols > 0, "slab_reduce: bias / ld_out need n_cols");
    const int grid = (int)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs, S, slab_stride, bias,
                       n_cols, act, out, ld_out, numel);
    ERC_LAUNCH_CHECK("slab_reduce");
    return ERC_OK;
}

extern "C" int erc_slab_reduce_batched(const float* ws, float* dst, const int64_t* jobs, int n_jobs,
                                       int64_t max_numel, void* stream) {
    ERC_REQUIRE(ws && dst && jobs && n_jobs > 0 && max_numel > 0, "slab_reduce_batched: bad arguments");
    int gx = (int)((max_numel + 255) / 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(slab_reduce_batched_kernel, dim3(gx, n_jobs), dim3(256), 0, (hipStream_t)stream, ws, dst, jobs);
    ERC_LAUNCH_CHECK("slab_reduce_batched");
    return ERC_OK;
}
