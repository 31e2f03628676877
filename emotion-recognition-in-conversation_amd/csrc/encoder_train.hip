// K9, training half: the pieces the chained COGMEN variant (`transformer_out(encoder(x))` with the
// key-padding mask, instead of the reference's computed-and-discarded encoder, track_mm/cogmen.py:145-147) needs on
// top of encoder.hip -- layer math contrib/nn.py:283-305 in training mode (4 dropout sites per layer) and its backward.
//
//   attention (fwd / bwd)  one workgroup per (dialogue, head), S <= 128 positions, head dim <= 256: wavefront w owns the
//                          query rows [16 w, 16 w + 16); q k^T, p v and the four backward products run on
//                          v_mfma_f32_16x16x32_bf16 with K / V / Q / dO staged through LDS in 96-column chunks
//                          (natural layout for products that contract over the head dim, transposed for those that
//                          contract over positions).  Nothing is saved by the forward: the backward recomputes the
//                          probabilities (and the dropout decisions from the counter-based generator).
//   add + LayerNorm        y = LN(a + dropout(b)); keeps the pre-norm sum and (mean, rstd) for the backward
//   LayerNorm backward     d(pre-norm sum) as fp32 (residual branch) and, through the dropout mask, as bf16 (the
//                          branch that feeds the dense products) + per-workgroup partial column sums for gamma / beta
//   transpose              [R, C] fp32 | bf16 -> bf16 [C, R_pad]: weight gradients are NT products over the R axis
//   column sums            bias gradients, two-stage and order-fixed (no float atomics)
#include "erc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f2bf(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// ------------------------------------------------------------------------------------------------ attention
constexpr int AT_S = 128;     // positions per sequence (8 row tiles of 16)
constexpr int AT_DC = 96;     // head-dim columns per staged chunk
constexpr int AT_KP = 104;    // LDS pitch of a natural chunk  [position][96 + 8]
constexpr int AT_PP = 136;    // LDS pitch of [.][128 + 8] arrays (transposed chunks, probabilities)
constexpr int AT_A_ELEMS = AT_S * AT_KP;   // 13 312 >= 96 * 136
constexpr int AT_P_ELEMS = AT_S * AT_PP;

struct AttnP {
    const unsigned short* qkv;    // [n_seq * S, 3 D]
    const unsigned short* dout;   // [n_seq * S, D]      (backward)
    unsigned short* out;          // [n_seq * S, D]      (forward)
    unsigned short* dqkv;         // [n_seq * S, 3 D]    (backward)
    const int64_t* lengths;       // valid keys per sequence, or null (no padding mask)
    const uint64_t* rng;          // {offset, seed}
    uint64_t rng_stream;
    int S, D, heads, hd;
    float scale, drop_p;
};

// reduce over the 16 lanes of a 16-lane row (the lanes that share l >> 4)
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
    v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 1, 64));
    return v;
}

// sA[pos][c] = src[pos, c0 + c] (c < 96), zero outside [0, S) x [0, hd).  Loads in batches of 8 per thread, then the 8
// LDS stores: one element per loop iteration is one exposed L2 round trip per iteration (24 per chunk).
__device__ __forceinline__ void stage_natural(unsigned short* sA, const unsigned short* src, int ld, int S, int hd, int c0, int n_rows) {
    const int total = n_rows * AT_DC;
    for (int base = threadIdx.x; base < total; base += 512 * 8) {
        unsigned short v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = min(base + 512 * u, total - 1), j = idx / AT_DC, c = idx - j * AT_DC;
            v[u] = src[(int64_t)min(j, S - 1) * ld + min(c0 + c, hd - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + 512 * u, j = idx / AT_DC, c = idx - j * AT_DC;
            const bool ok = j < S && c0 + c < hd;
            if (idx < total) sA[j * AT_KP + c] = v[u] & (ok ? (unsigned short)0xffffu : (unsigned short)0);
        }
    }
}
// sA[c][pos] = src[pos, c0 + c] for all 128 positions (zero outside): the contraction runs over positions
__device__ __forceinline__ void stage_transposed(unsigned short* sA, const unsigned short* src, int ld, int S, int hd, int c0) {
    constexpr int total = AT_S * AT_DC;      // 24 elements per thread
    for (int base = threadIdx.x; base < total; base += 512 * 8) {
        unsigned short v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + 512 * u, j = idx / AT_DC, c = idx - j * AT_DC;
            v[u] = src[(int64_t)min(j, S - 1) * ld + min(c0 + c, hd - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + 512 * u, j = idx / AT_DC, c = idx - j * AT_DC;
            const bool ok = j < S && c0 + c < hd;
            sA[c * AT_PP + j] = v[u] & (ok ? (unsigned short)0xffffu : (unsigned short)0);
        }
    }
}
// acc[jt] += F(rows of tile w, head dim) * X(positions, head dim)^T, both operands staged chunk by chunk in natural
// layout: F in the (not yet used) probability region, X in the chunk region
__device__ __forceinline__ void rows_times_xt(f32x4 (&acc)[8], unsigned short* sF, unsigned short* sA, const unsigned short* f, int ldf,
                                              const unsigned short* x, int ldx, int S, int hd, int nt, int nkb, bool active,
                                              int w, int r, int g) {
    for (int c0 = 0; c0 < hd; c0 += AT_DC) {
        __syncthreads();
        stage_natural(sF, f, ldf, S, hd, c0, 16 * nt);
        stage_natural(sA, x, ldx, S, hd, c0, AT_S);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                if (c0 / 32 + kk < nkb) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sF[(16 * w + r) * AT_KP + 32 * kk + 8 * g]);
#pragma unroll
                    for (int jt = 0; jt < 8; ++jt)
                        {   // all 8 key tiles: a condition on jt here makes hipcc spill the accumulators
                            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&sA[(16 * jt + r) * AT_KP + 32 * kk + 8 * g]);
                            acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[jt], 0, 0, 0);
                        }
                }
            }
        }
    }
}
// out[16 rows of tile w][head dim] = sP(rows of tile w, positions) * X(positions, head dim), X staged transposed;
// result rows 16 w + 4 g + q, columns c0 + 16 nt + r  ->  dst[row * ld + col] (bf16)
__device__ __forceinline__ void p_times_x(const unsigned short* sP, unsigned short* sA, const unsigned short* x, int ldx,
                                          unsigned short* dst, int ldd, int S, int hd, int nkj, bool active, int w, int r, int g,
                                          float out_scale) {
    for (int c0 = 0; c0 < hd; c0 += AT_DC) {
        __syncthreads();
        stage_transposed(sA, x, ldx, S, hd, c0);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int nt = 0; nt < AT_DC / 16; ++nt) {
                if (c0 + 16 * nt < hd) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kj = 0; kj < 4; ++kj) {   // all 128 positions: both operands are zero beyond S
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sP[(16 * w + r) * AT_PP + 32 * kj + 8 * g]);
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&sA[(16 * nt + r) * AT_PP + 32 * kj + 8 * g]);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
                    }
                    const int col = c0 + 16 * nt + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 16 * w + 4 * g + q;
                        if (row < S && col < hd) dst[(int64_t)row * ldd + col] = f2bf(acc[q] * out_scale);
                    }
                }
            }
        }
    }
}

// softmax over the keys of this wavefront's 16 query rows (scores in acc, C layout: register q of lane (r, g) = row
// 4 g + q, key 16 jt + r), then attention dropout.  p = probabilities, return = dropout decisions (bit 4 jt + q set: kept).
__device__ __forceinline__ unsigned softmax_rows(const f32x4 (&acc)[8], float (&p)[8][4], const AttnP& P, int b_h,
                                                 int n_keys, int nt, int w, int r, int g) {
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool kv = jt < nt && 16 * jt + r < n_keys;
            p[jt][q] = kv ? acc[jt][q] * P.scale : -INFINITY;
            mx[q] = fmaxf(mx[q], p[jt][q]);
        }
    float den[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mx[q] = row16_max(mx[q]), den[q] = 0.f;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p[jt][q] = expf(p[jt][q] - mx[q]);      // masked keys: exp(-inf) = 0
            den[q] += p[jt][q];
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) den[q] = 1.0f / row16_sum(den[q]);
    unsigned keep = 0xffffffffu;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) p[jt][q] *= den[q];
    if (P.drop_p > 0.f) {
        const uint64_t off = P.rng[0], seed = P.rng[1] ^ P.rng_stream;
        keep = 0u;
        for (int jt = 0; jt < 8; ++jt)      // not unrolled: 32 splitmix chains would only add register pressure
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t i = 16 * w + 4 * g + q, j = 16 * jt + r;
                const float u = erc_uniform(seed, off, ((uint64_t)b_h * (uint64_t)P.S + i) * (uint64_t)P.S + j);
                keep |= (u >= P.drop_p ? 1u : 0u) << (4 * jt + q);
            }
    }
    return keep;
}
// sP[row 16 w + 4 g + q][key 16 jt + r] = v (wave-private rows)
__device__ __forceinline__ void store_rows(unsigned short* sP, const float (&v)[8][4], int w, int r, int g) {
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) sP[(16 * w + 4 * g + q) * AT_PP + 16 * jt + r] = f2bf(v[jt][q]);
}
// sP[key 16 jt + r][row 16 w + 4 g + q] = v: 4 consecutive rows per 8-byte store
__device__ __forceinline__ void store_cols(unsigned short* sP, const float (&v)[8][4], int w, int r, int g) {
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
        const bf16x4 pk = {(short)f2bf(v[jt][0]), (short)f2bf(v[jt][1]), (short)f2bf(v[jt][2]), (short)f2bf(v[jt][3])};
        *reinterpret_cast<bf16x4*>(&sP[(16 * jt + r) * AT_PP + 16 * w + 4 * g]) = pk;
    }
}

__global__ __launch_bounds__(512) void enc_attn_train_fwd_kernel(AttnP P) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[AT_A_ELEMS], sP[AT_P_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int b_h = blockIdx.x, b = b_h / P.heads, h = b_h - b * P.heads;
    const int S = P.S, hd = P.hd, ld = 3 * P.D;
    const int nt = (S + 15) / 16, nkb = (hd + 31) / 32, nkj = (nt + 1) / 2;
    const bool active = w < nt;
    const unsigned short* base = P.qkv + (int64_t)b * S * ld + h * hd;
    const int n_keys = P.lengths ? (int)min((int64_t)S, P.lengths[b]) : S;

    f32x4 acc[8];
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) acc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    rows_times_xt(acc, sP, sA, base, ld, base + P.D, ld, S, hd, nt, nkb, active, w, r, g);

    float p[8][4];
    const unsigned keep = softmax_rows(acc, p, P, b_h, n_keys, nt, w, r, g);
    const float ks = 1.0f / (1.0f - P.drop_p);
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) p[jt][q] = (active && ((keep >> (4 * jt + q)) & 1u)) ? p[jt][q] * ks : 0.f;
    __syncthreads();      // every wavefront is done reading its query rows out of sP
    store_rows(sP, p, w, r, g);
    p_times_x(sP, sA, base + 2 * P.D, ld, P.out + (int64_t)b * S * P.D + h * hd, P.D, S, hd, nkj, active, w, r, g, 1.f);
}

__global__ __launch_bounds__(512) void enc_attn_bwd_kernel(AttnP P) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[AT_A_ELEMS], sP[AT_P_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int b_h = blockIdx.x, b = b_h / P.heads, h = b_h - b * P.heads;
    const int S = P.S, hd = P.hd, ld = 3 * P.D;
    const int nt = (S + 15) / 16, nkb = (hd + 31) / 32, nkj = (nt + 1) / 2;
    const bool active = w < nt;
    const unsigned short* base = P.qkv + (int64_t)b * S * ld + h * hd;
    const unsigned short* dob = P.dout + (int64_t)b * S * P.D + h * hd;
    unsigned short* dbase = P.dqkv + (int64_t)b * S * ld + h * hd;
    const int n_keys = P.lengths ? (int)min((int64_t)S, P.lengths[b]) : S;

    float p[8][4], ds[8][4];
    {
        f32x4 acc[8];
        // scores -> probabilities
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) acc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rows_times_xt(acc, sP, sA, base, ld, base + P.D, ld, S, hd, nt, nkb, active, w, r, g);
        const unsigned keep = softmax_rows(acc, p, P, b_h, n_keys, nt, w, r, g);
        const float ks = 1.0f / (1.0f - P.drop_p);
        // d(dropped probabilities) = dO V^T
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) acc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        rows_times_xt(acc, sP, sA, dob, P.D, base + 2 * P.D, ld, S, hd, nt, nkb, active, w, r, g);
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float kf = ((keep >> (4 * jt + q)) & 1u) ? ks : 0.f;
                ds[jt][q] = acc[jt][q] * kf;                  // dP
                rs[q] += p[jt][q] * ds[jt][q];
                acc[jt][q] = kf;
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) rs[q] = row16_sum(rs[q]);
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ds[jt][q] = active ? p[jt][q] * (ds[jt][q] - rs[q]) * P.scale : 0.f;   // d(q k^T), scale folded in
                p[jt][q] = active ? p[jt][q] * acc[jt][q] : 0.f;                      // dropped probabilities
            }
    }
    // dV = Pd^T dO
    __syncthreads();
    store_cols(sP, p, w, r, g);
    p_times_x(sP, sA, dob, P.D, dbase + 2 * P.D, ld, S, hd, nkj, active, w, r, g, 1.f);
    // dK = dS^T Q
    __syncthreads();
    store_cols(sP, ds, w, r, g);
    p_times_x(sP, sA, base, ld, dbase + P.D, ld, S, hd, nkj, active, w, r, g, 1.f);
    // dQ = dS K
    __syncthreads();
    store_rows(sP, ds, w, r, g);
    p_times_x(sP, sA, base + P.D, ld, dbase, ld, S, hd, nkj, active, w, r, g, 1.f);
}

// ------------------------------------------------------------------------------------------------ LayerNorm
struct LnP {
    const float* a; const float* b;               // forward: y = LN(a + dropout(b))
    const float* gamma; const float* beta;
    float* yf; unsigned short* yh;
    float* saved_s; float* stats;                 // pre-norm sum [n_rows, D]; mean [n_rows] | rstd [n_rows]
    const float* dy_a; const int* dy_a_map; const float* dy_b;   // backward: dy = dy_a[map(row)] + dy_b
    float* ds; unsigned short* db; float* partial;               // partial: [gridDim.x][2][D]
    const uint64_t* rng; uint64_t rng_stream;
    int D, n_rows;
    float eps, drop_p;
};
constexpr int LN_MAXD = 2048;   // columns per lane NC = 12 | 24 | 32 (D <= 768 | 1536 | 2048)

template <int LN_MAXC>
__global__ __launch_bounds__(256) void enc_add_ln_train_kernel(LnP P) {
    const int lane = threadIdx.x & 63, row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P.n_rows) return;
    const int D = P.D;
    const float ks = 1.0f / (1.0f - P.drop_p);
    uint64_t off = 0, seed = 0;
    if (P.drop_p > 0.f) off = P.rng[0], seed = P.rng[1] ^ P.rng_stream;
    float v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < LN_MAXC; ++u) {
        const int c = lane + 64 * u;
        const int64_t at = (int64_t)row * D + min(c, D - 1);
        float bv = P.b[at];
        if (P.drop_p > 0.f) bv = erc_uniform(seed, off, (uint64_t)at) >= P.drop_p ? bv * ks : 0.f;
        v[u] = c < D ? P.a[at] + bv : 0.f;
        s += v[u];
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < LN_MAXC; ++u) {
        const float d = lane + 64 * u < D ? v[u] - mean : 0.f;
        ss += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + P.eps);
    if (lane == 0) P.stats[row] = mean, P.stats[P.n_rows + row] = rstd;
#pragma unroll
    for (int u = 0; u < LN_MAXC; ++u) {
        const int c = lane + 64 * u;
        if (c < D) {
            const float y = (v[u] - mean) * rstd * P.gamma[c] + P.beta[c];
            P.saved_s[(int64_t)row * D + c] = v[u];
            P.yf[(int64_t)row * D + c] = y;
            P.yh[(int64_t)row * D + c] = f2bf(y);
        }
    }
}

// One wavefront per row at a time, two passes over the row (8 columns per lane in flight, no per-row arrays: 65 VGPRs);
// the gamma / beta column sums of a wavefront's rows accumulate in its private LDS strip and the four strips are added
// in a fixed order at the end.  HAS_B is a template parameter: with `if (dy_b) dy += dy_b[..]` in the column loop hipcc
// branches around the load and waits for it on the spot (74 us per launch instead of 30).
template <int LN_MAXC, bool HAS_B>
__global__ __launch_bounds__(256) void enc_ln_bwd_kernel(LnP P) {
    __shared__ float sAcc[4][2][LN_MAXC * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, D = P.D;
    const float ks = 1.0f / (1.0f - P.drop_p);
    uint64_t off = 0, seed = 0;
    if (P.drop_p > 0.f) off = P.rng[0], seed = P.rng[1] ^ P.rng_stream;
#pragma unroll
    for (int u = 0; u < LN_MAXC; ++u) sAcc[w][0][lane + 64 * u] = 0.f, sAcc[w][1][lane + 64 * u] = 0.f;
    for (int row = (int)blockIdx.x * 4 + w; row < P.n_rows; row += (int)gridDim.x * 4) {
        const float mean = P.stats[row], rstd = P.stats[P.n_rows + row];
        const int src = P.dy_a_map ? P.dy_a_map[row] : row;
        const float am = src >= 0 ? 1.f : 0.f;
        const float* dya = P.dy_a + (int64_t)max(src, 0) * D;
        const float* dyb = HAS_B ? P.dy_b + (int64_t)row * D : dya;
        const float* srow = P.saved_s + (int64_t)row * D;
        float m1 = 0.f, m2 = 0.f;
#pragma unroll 8
        for (int u = 0; u < LN_MAXC; ++u) {
            const int c = lane + 64 * u, cc = min(c, D - 1);
            const float cm = c < D ? 1.f : 0.f;
            float dy = dya[cc] * am;
            if (HAS_B) dy += dyb[cc];
            dy *= cm;
            const float xh = (srow[cc] - mean) * rstd * cm;
            sAcc[w][0][c] += dy * xh;
            sAcc[w][1][c] += dy;
            const float gy = dy * P.gamma[cc];
            m1 += gy;
            m2 += gy * xh;
        }
        m1 = wave_sum(m1) / (float)D;
        m2 = wave_sum(m2) / (float)D;
#pragma unroll 8
        for (int u = 0; u < LN_MAXC; ++u) {
            const int c = lane + 64 * u, cc = min(c, D - 1);
            float dy = dya[cc] * am;
            if (HAS_B) dy += dyb[cc];
            const float xh = (srow[cc] - mean) * rstd;
            const float d = rstd * (dy * P.gamma[cc] - m1 - xh * m2);
            if (c < D) {
                const int64_t at = (int64_t)row * D + c;
                P.ds[at] = d;
                float db = d;
                if (P.drop_p > 0.f) db = erc_uniform(seed, off, (uint64_t)at) >= P.drop_p ? d * ks : 0.f;
                P.db[at] = f2bf(db);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        P.partial[((int64_t)blockIdx.x * 2 + 0) * D + c] = (sAcc[0][0][c] + sAcc[1][0][c]) + (sAcc[2][0][c] + sAcc[3][0][c]);
        P.partial[((int64_t)blockIdx.x * 2 + 1) * D + c] = (sAcc[0][1][c] + sAcc[1][1][c]) + (sAcc[2][1][c] + sAcc[3][1][c]);
    }
}

// ------------------------------------------------------------------------------------------------ transpose / column sums
// YT[c][r] = bf16(X[r][c]) for r < R (zero up to the pitch), optional plain bf16 copy; 64 x 64 tiles through LDS
template <bool SRC_F32>
__global__ __launch_bounds__(256) void enc_transpose_kernel(const void* __restrict__ Xv, int ldx, int R, int C,
                                                            unsigned short* __restrict__ YT, int ldyt,
                                                            unsigned short* __restrict__ Pl, int ldp) {
    __shared__ unsigned short sT[64][68];
    const int tid = threadIdx.x, r0 = (int)blockIdx.y * 64, c0 = (int)blockIdx.x * 64;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = 16 * pass + (tid >> 4), col = (tid & 15) * 4;
        const int gr = min(r0 + row, R - 1);
        unsigned short v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gc = min(c0 + col + e, C - 1);
            const bool ok = r0 + row < R && c0 + col + e < C;
            unsigned short x;
            if (SRC_F32) x = f2bf(reinterpret_cast<const float*>(Xv)[(int64_t)gr * ldx + gc]);
            else x = reinterpret_cast<const unsigned short*>(Xv)[(int64_t)gr * ldx + gc];
            v[e] = x & (ok ? (unsigned short)0xffffu : (unsigned short)0);
            sT[row][col + e] = v[e];
            if (Pl && ok) Pl[(int64_t)(r0 + row) * ldp + c0 + col + e] = v[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int cc = 16 * pass + (tid >> 4), rr = (tid & 15) * 4;
        if (c0 + cc < C && r0 + rr < ldyt) {
            const bf16x4 pk = {(short)sT[rr][cc], (short)sT[rr + 1][cc], (short)sT[rr + 2][cc], (short)sT[rr + 3][cc]};
            *reinterpret_cast<bf16x4*>(&YT[(int64_t)(c0 + cc) * ldyt + r0 + rr]) = pk;
        }
    }
}

// out[split][c] = sum over this split's rows of X[r][c]; 4 row lanes x 64 columns per workgroup, 8 loads in flight
template <bool SRC_BF16>
__global__ __launch_bounds__(256) void enc_colsum_kernel(const void* __restrict__ Xv, int ldx, int R, int C, int rows_per_split,
                                                         float* __restrict__ out) {
    __shared__ float sR[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = (int)blockIdx.x * 64 + cl, cc = min(c, C - 1);
    const int r_begin = (int)blockIdx.y * rows_per_split, r_end = min(R, r_begin + rows_per_split);
    float acc = 0.f;
    for (int rb = r_begin + rl; rb < r_end; rb += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = rb + 4 * u;
            const int64_t at = (int64_t)min(rr, R - 1) * ldx + cc;
            const float x = SRC_BF16 ? bf2f(reinterpret_cast<const unsigned short*>(Xv)[at]) : reinterpret_cast<const float*>(Xv)[at];
            v[u] = x * (rr < r_end ? 1.f : 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    sR[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C) out[(int64_t)blockIdx.y * C + c] = (sR[0][cl] + sR[1][cl]) + (sR[2][cl] + sR[3][cl]);
}

__global__ __launch_bounds__(256) void enc_inverse_rows_kernel(const int* __restrict__ node_row, int N, int* __restrict__ inv) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i < N) inv[node_row[i]] = i;
}

constexpr int CS_SPLITS = 32;

}  // namespace

extern "C" int erc_enc_attention_train(const void* qkv, int n_seq, int S, int D, int heads, const int64_t* lengths, float drop_p,
                                       const uint64_t* rng_state, uint64_t rng_stream, void* out, void* stream) {
    ERC_REQUIRE(qkv && out && n_seq > 0 && S > 0 && S <= AT_S && heads > 0 && D % heads == 0 && D / heads <= 256,
                "enc_attention_train: bad arguments (S <= 128, head dim <= 256)");
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state), "enc_attention_train: drop_p=%f", (double)drop_p);
    AttnP p{};
    p.qkv = (const unsigned short*)qkv, p.out = (unsigned short*)out, p.lengths = lengths, p.rng = rng_state, p.rng_stream = rng_stream;
    p.S = S, p.D = D, p.heads = heads, p.hd = D / heads, p.scale = 1.0f / sqrtf((float)(D / heads)), p.drop_p = drop_p;
    hipLaunchKernelGGL(enc_attn_train_fwd_kernel, dim3(n_seq * heads), dim3(512), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("enc_attention_train");
    return ERC_OK;
}

extern "C" int erc_enc_attention_bwd(const void* qkv, const void* dout, int n_seq, int S, int D, int heads, const int64_t* lengths,
                                     float drop_p, const uint64_t* rng_state, uint64_t rng_stream, void* dqkv, void* stream) {
    ERC_REQUIRE(qkv && dout && dqkv && n_seq > 0 && S > 0 && S <= AT_S && heads > 0 && D % heads == 0 && D / heads <= 256,
                "enc_attention_bwd: bad arguments (S <= 128, head dim <= 256)");
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state), "enc_attention_bwd: drop_p=%f", (double)drop_p);
    AttnP p{};
    p.qkv = (const unsigned short*)qkv, p.dout = (const unsigned short*)dout, p.dqkv = (unsigned short*)dqkv, p.lengths = lengths;
    p.rng = rng_state, p.rng_stream = rng_stream;
    p.S = S, p.D = D, p.heads = heads, p.hd = D / heads, p.scale = 1.0f / sqrtf((float)(D / heads)), p.drop_p = drop_p;
    hipLaunchKernelGGL(enc_attn_bwd_kernel, dim3(n_seq * heads), dim3(512), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("enc_attention_bwd");
    return ERC_OK;
}

extern "C" int erc_enc_add_layernorm_train(const float* a, const float* b, int D, int n_rows, const float* gamma, const float* beta,
                                           float eps, float drop_p, const uint64_t* rng_state, uint64_t rng_stream, float* y_f32,
                                           void* y_bf16, float* saved_sum, float* saved_stats, void* stream) {
    ERC_REQUIRE(a && b && gamma && beta && y_f32 && y_bf16 && saved_sum && saved_stats && n_rows > 0 && D > 0 && D <= LN_MAXD,
                "enc_add_layernorm_train: bad arguments");
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state), "enc_add_layernorm_train: drop_p=%f", (double)drop_p);
    LnP p{};
    p.a = a, p.b = b, p.gamma = gamma, p.beta = beta, p.yf = y_f32, p.yh = (unsigned short*)y_bf16, p.saved_s = saved_sum;
    p.stats = saved_stats, p.rng = rng_state, p.rng_stream = rng_stream, p.D = D, p.n_rows = n_rows, p.eps = eps, p.drop_p = drop_p;
    const dim3 grid(erc_cdiv(n_rows, 4));
    if (D <= 768) hipLaunchKernelGGL(enc_add_ln_train_kernel<12>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (D <= 1536) hipLaunchKernelGGL(enc_add_ln_train_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(enc_add_ln_train_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("enc_add_layernorm_train");
    return ERC_OK;
}

extern "C" int erc_enc_layernorm_bwd_blocks(int n_rows) { return n_rows < 2048 ? (n_rows + 3) / 4 : 512; }

extern "C" int erc_enc_layernorm_bwd(const float* dy_a, const int32_t* dy_a_map, const float* dy_b, const float* saved_sum,
                                     const float* saved_stats, const float* gamma, int D, int n_rows, float drop_p,
                                     const uint64_t* rng_state, uint64_t rng_stream, float* ds, void* db_bf16, float* partial,
                                     void* stream) {
    ERC_REQUIRE(dy_a && saved_sum && saved_stats && gamma && ds && db_bf16 && partial && n_rows > 0 && D > 0 && D <= LN_MAXD,
                "enc_layernorm_bwd: bad arguments");
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state), "enc_layernorm_bwd: drop_p=%f", (double)drop_p);
    LnP p{};
    p.dy_a = dy_a, p.dy_a_map = dy_a_map, p.dy_b = dy_b, p.saved_s = const_cast<float*>(saved_sum);
    p.stats = const_cast<float*>(saved_stats), p.gamma = gamma, p.ds = ds, p.db = (unsigned short*)db_bf16, p.partial = partial;
    p.rng = rng_state, p.rng_stream = rng_stream, p.D = D, p.n_rows = n_rows, p.drop_p = drop_p;
    const dim3 grid(erc_enc_layernorm_bwd_blocks(n_rows));
    auto launch = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, p); };
    if (dy_b) {
        if (D <= 768) launch(enc_ln_bwd_kernel<12, true>);
        else if (D <= 1536) launch(enc_ln_bwd_kernel<24, true>);
        else launch(enc_ln_bwd_kernel<32, true>);
    } else {
        if (D <= 768) launch(enc_ln_bwd_kernel<12, false>);
        else if (D <= 1536) launch(enc_ln_bwd_kernel<24, false>);
        else launch(enc_ln_bwd_kernel<32, false>);
    }
    ERC_LAUNCH_CHECK("enc_layernorm_bwd");
    return ERC_OK;
}

extern "C" int erc_enc_transpose_bf16(const void* X, int x_is_f32, int ldx, int R, int C, void* YT, int ldyt, void* plain, int ldp,
                                      void* stream) {
    ERC_REQUIRE(X && YT && R > 0 && C > 0 && ldx >= C && ldyt >= R && ldyt % 4 == 0 && ((uintptr_t)YT & 7) == 0 && (!plain || ldp >= C),
                "enc_transpose_bf16: bad arguments (transposed pitch: multiple of 4, >= R)");
    const dim3 grid(erc_cdiv(C, 64), erc_cdiv(ldyt, 64));
    if (x_is_f32)
        hipLaunchKernelGGL(enc_transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, R, C, (unsigned short*)YT,
                           ldyt, (unsigned short*)plain, ldp);
    else
        hipLaunchKernelGGL(enc_transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, X, ldx, R, C, (unsigned short*)YT,
                           ldyt, (unsigned short*)plain, ldp);
    ERC_LAUNCH_CHECK("enc_transpose_bf16");
    return ERC_OK;
}

extern "C" int64_t erc_enc_colsum_ws_floats(int C) { return (int64_t)CS_SPLITS * C; }

extern "C" int erc_enc_colsum(const void* X, int x_is_bf16, int ldx, int R, int C, float* out, float* ws, void* stream) {
    ERC_REQUIRE(X && out && ws && R > 0 && C > 0 && ldx >= C, "enc_colsum: bad arguments");
    const int splits = R >= 4 * CS_SPLITS ? CS_SPLITS : 1;
    const int rps = (R + splits - 1) / splits;
    float* first = splits > 1 ? ws : out;
    if (x_is_bf16)
        hipLaunchKernelGGL(enc_colsum_kernel<true>, dim3(erc_cdiv(C, 64), splits), dim3(256), 0, (hipStream_t)stream, X, ldx, R, C, rps, first);
    else
        hipLaunchKernelGGL(enc_colsum_kernel<false>, dim3(erc_cdiv(C, 64), splits), dim3(256), 0, (hipStream_t)stream, X, ldx, R, C, rps, first);
    if (splits > 1)
        hipLaunchKernelGGL(enc_colsum_kernel<false>, dim3(erc_cdiv(C, 64), 1), dim3(256), 0, (hipStream_t)stream, (const void*)ws, C,
                           splits, C, splits, out);
    ERC_LAUNCH_CHECK("enc_colsum");
    return ERC_OK;
}

extern "C" int erc_enc_inverse_rows(const int32_t* node_row, int N, int32_t* inv, int n_rows, void* stream) {
    ERC_REQUIRE(node_row && inv && N > 0 && n_rows >= N, "enc_inverse_rows: bad arguments");
    hipError_t e = hipMemsetAsync(inv, 0xff, (size_t)n_rows * sizeof(int32_t), (hipStream_t)stream);
    ERC_REQUIRE(e == hipSuccess, "enc_inverse_rows: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(enc_inverse_rows_kernel, dim3(erc_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, node_row, N, inv);
    ERC_LAUNCH_CHECK("enc_inverse_rows");
    return ERC_OK;
}
