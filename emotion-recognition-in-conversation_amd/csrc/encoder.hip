// K9: the Transformer encoder block of COGMEN's `rnn.0` (track_mm/cogmen.py:94-102, layer math contrib/nn.py:283-305:
// post-norm, ReLU, ffn 2048, batch_first, NO padding mask).  In the reference its output is discarded
// (cogmen.py:146-147), so this block is only run in the faithful-cost mode and by the tests;
// inference-mode math (dropout layers are identities here: the result is thrown away, the arithmetic is the same).
//
//   qkv = x Win^T + b ; per (dialogue, head): softmax(q k^T / sqrt(hd)) v over the T padded positions ;
//   x1 = LN1(x + attn Wo^T + bo) ; x2 = LN2(x1 + relu(x1 W1^T + b1) W2^T + b2)
//
// 97 % of the 95 GFLOP per layer are the four dense products: bf16 operands, fp32 accumulate on
// v_mfma_f32_16x16x32_bf16, 128 x 128 x 32 workgroup tiles staged through LDS (rows padded to 80 bytes: the 16-byte
// fragment reads of the 16 lanes r then fall into 16 distinct bank groups), double buffered, 8 wavefronts of 32 x 64.
#include <stdlib.h>

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

struct EgP0 { static constexpr int value = 0; };
struct EgP1 { static constexpr int value = 1; };
constexpr int EG_BM = 128, EG_BN = 128, EG_BK = 64;
constexpr int EG_PITCH = 64;   // bf16 elements per LDS row: unpadded, chunk positions XOR-swizzled (see the kernel)
constexpr int EG_TILE = 128 * EG_PITCH;            // elements of one operand tile
constexpr int EG_LDS_BYTES = 4 * EG_TILE * 2;     // A, B x 2 buffers = 65 536 bytes (dynamic)
typedef bf16x8 bf16x8_a8 __attribute__((aligned(8)));

// epilogue extras of the training path: EPI 1 = dropout after the (optional) ReLU, EPI 2 = multiply by
// (mask_src[row, col] != 0) * scale: the backward of ReLU + dropout, read off the stored forward output
struct EgEpi {
    const unsigned short* mask_src;
    int ld_mask;
    float scale, drop_p;
    const uint64_t* rng;
    uint64_t rng_stream;
};

// C[M,N] = A[M,K] W[N,K]^T + bias (+ReLU); A, W bf16 with K contiguous (K, lda, ldw multiples of 4, K >= 8); C fp32 and /
// or bf16.  128 x 128 x 64 workgroup tiles, 8 wavefronts (4 x 2) of 32 x 64, LDS double buffer (2 x 32 KB, dynamic).
//  * operand rows are only 8-byte aligned (K = 1380): the 16-byte global loads are issued on 8-byte-aligned addresses
//    (the hardware takes dword-aligned dwordx4 loads); a chunk that straddles K is loaded 4 elements early and
//    shifted when it is stored to LDS, so nothing past a row's end is ever read;
//  * straight-line pipeline, prefetch distance 2 through two register sets: every step issues its loads and its LDS
//    store unconditionally (tiles past the end are clamped to a valid address and zeroed at the store; the k-block
//    count is rounded up to even).  Masking the loaded value, or branching around the prefetch, makes hipcc wait for /
//    drain the loads on the spot;
//  * 4 wavefronts of 64 x 64 (a third fewer LDS fragment reads per flop) were 10 - 35 % slower on every encoder shape:
//    one wavefront per SIMD and workgroup does not cover the barrier per k-step.
template <int EPI>
__global__ __launch_bounds__(512, 4) void enc_gemm_kernel(const unsigned short* __restrict__ A, int lda,
                                                       const unsigned short* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias, float* __restrict__ Cf,
                                                       unsigned short* __restrict__ Ch, int ldc, int M, int N, int K, int relu,
                                                       EgEpi ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned short eg_lds[];
    unsigned short* sA[2] = {eg_lds, eg_lds + EG_TILE};
    unsigned short* sB[2] = {eg_lds + 2 * EG_TILE, eg_lds + 3 * EG_TILE};
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    // XCD-aware tile order: workgroup id b runs on XCD b % 8 (each XCD has its own L2).  XCD x takes the contiguous run
    // of tiles [x * per, (x + 1) * per) in row-major (m, n) order, so that the few A row-tiles it touches and W stay in
    // its L2 instead of every XCD streaming all of A.
    const int tiles_n = (N + EG_BN - 1) / EG_BN, n_tiles = tiles_n * ((M + EG_BM - 1) / EG_BM);
    const int per = (n_tiles + 7) / 8;
    const int tile = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (tile >= n_tiles || ((int)blockIdx.x >> 3) >= per) return;
    const int m0 = (tile / tiles_n) * EG_BM, n0 = (tile % tiles_n) * EG_BN;
    const int nkb = (K + EG_BK - 1) / EG_BK;

    // this thread's two 16-byte chunks of an operand tile (128 rows x 8 chunks of 8 bf16): rows 16 w + sub (+ 8), chunk
    // lane & 7.  LDS rows are unpadded (128 bytes); chunk c of row q lives at chunk position c ^ ((q >> 1) & 7).  With
    // that swizzle both the 16-byte fragment reads (lanes r = row, g = chunk) and the tile stores are bank-conflict
    // free; a padded pitch of 144 bytes measured 37 % of the LDS cycles as bank conflicts (SQ_LDS_BANK_CONFLICT /
    // SQ_LDS_IDX_ACTIVE) -- ds_read_b128 does not serve its 64 lanes in four groups of 16 consecutive lanes.
    const int sub = lane >> 3, ck = (lane & 7) * 8;
    const int crow[2] = {16 * w + sub, 16 * w + sub + 8};
    const int cpos[2] = {8 * ((lane & 7) ^ ((crow[0] >> 1) & 7)), 8 * ((lane & 7) ^ ((crow[1] >> 1) & 7))};
    const unsigned short* arow[2] = {A + (int64_t)min(m0 + crow[0], M - 1) * lda, A + (int64_t)min(m0 + crow[1], M - 1) * lda};
    const unsigned short* brow[2] = {W + (int64_t)min(n0 + crow[0], N - 1) * ldw, W + (int64_t)min(n0 + crow[1], N - 1) * ldw};
    bf16x8 ra[2][2], rb[2][2];
    auto gload = [&](int kb, bf16x8 (&qa)[2], bf16x8 (&qb)[2]) {
        const int k = kb * EG_BK + ck;
        const int kcl = k + 8 <= K ? k : (k < K ? K - 8 : 0);      // K % 4 == 0: a chunk is whole, half (tail) or absent
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            qa[u] = *reinterpret_cast<const bf16x8_a8*>(arow[u] + kcl);
            qb[u] = *reinterpret_cast<const bf16x8_a8*>(brow[u] + kcl);
        }
    };
    auto lstore = [&](int buf, int kb, const bf16x8 (&qa)[2], const bf16x8 (&qb)[2]) {
        const int k = kb * EG_BK + ck;
        const bool full = k + 8 <= K, tail = !full && k < K;       // tail: the valid 4 elements sit in the upper half
        const short mf = full ? (short)-1 : (short)0, mt = tail ? (short)-1 : (short)0;
        const bf16x8 m_lo = {mf, mf, mf, mf, mf, mf, mf, mf};
        const bf16x8 m_tl = {mt, mt, mt, mt, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 sa = __builtin_shufflevector(qa[u], qa[u], 4, 5, 6, 7, 0, 1, 2, 3);
            const bf16x8 sb = __builtin_shufflevector(qb[u], qb[u], 4, 5, 6, 7, 0, 1, 2, 3);
            *reinterpret_cast<bf16x8*>(&sA[buf][crow[u] * EG_PITCH + cpos[u]]) = (qa[u] & m_lo) | (sa & m_tl);
            *reinterpret_cast<bf16x8*>(&sB[buf][crow[u] * EG_PITCH + cpos[u]]) = (qb[u] & m_lo) | (sb & m_tl);
        }
    };
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nkb2 = (nkb + 1) & ~1;
    gload(0, ra[0], rb[0]);
    gload(1, ra[1], rb[1]);
    lstore(0, 0, ra[0], rb[0]);
    __syncthreads();
    auto step = [&](const int kb, auto parity) {
        constexpr int PAR = decltype(parity)::value;     // kb & 1 as a constant: register sets are indexed statically
        gload(kb + 2, ra[PAR], rb[PAR]);
        __builtin_amdgcn_sched_barrier(0);               // keep the prefetch at the top of the step
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(&sA[PAR][(wm * 32 + 16 * i + r) * EG_PITCH + 8 * ((4 * s2 + g) ^ ((r >> 1) & 7))]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[j] = *reinterpret_cast<const bf16x8*>(&sB[PAR][(wn * 64 + 16 * j + r) * EG_PITCH + 8 * ((4 * s2 + g) ^ ((r >> 1) & 7))]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        lstore(PAR ^ 1, kb + 1, ra[PAR ^ 1], rb[PAR ^ 1]);
        __syncthreads();
    };
    for (int kb = 0; kb < nkb2; kb += 2) {
        step(kb, EgP0{});
        step(kb + 1, EgP1{});
    }
    uint64_t rng_off = 0, rng_seed = 0;
    if (EPI == 1) rng_off = ep.rng[0], rng_seed = ep.rng[1] ^ ep.rng_stream;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + 16 * j + r;
        const float bv = bias ? bias[min(col, N - 1)] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + wm * 32 + 16 * i + 4 * g + q;
                if (row < M && col < N) {
                    float v = acc[i][j][q] + bv;
                    if (relu) v = fmaxf(v, 0.f);
                    if (EPI == 1)
                        v = erc_uniform(rng_seed, rng_off, (uint64_t)row * (uint64_t)N + col) >= ep.drop_p ? v * ep.scale : 0.f;
                    if (EPI == 2) v = ep.mask_src[(int64_t)row * ep.ld_mask + col] != 0 ? v * ep.scale : 0.f;
                    if (Cf) Cf[(int64_t)row * ldc + col] = v;
                    if (Ch) Ch[(int64_t)row * ldc + col] = f2bf(v);
                }
            }
    }
}

// Attention of one (sequence, head) per workgroup slice: one wavefront per query row, keys in batches of 16 (butterfly
// reduction of the 16 dot products), online softmax.  qkv bf16 [n_seq * S, 3 D] (q | k | v), out bf16 [n_seq * S, D].
__device__ __forceinline__ float butterfly16_sum(const float (&a)[16], int lane) {
    float b[8], c[4], d[2];
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (h5 ? a[8 + j] : a[j]) + __shfl_xor(h5 ? a[j] : a[8 + j], 32, 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = (h4 ? b[4 + j] : b[j]) + __shfl_xor(h4 ? b[j] : b[4 + j], 16, 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) d[j] = (h3 ? c[2 + j] : c[j]) + __shfl_xor(h3 ? c[j] : c[2 + j], 8, 64);
    float e = (h2 ? d[1] : d[0]) + __shfl_xor(h2 ? d[0] : d[1], 4, 64);
    e += __shfl_xor(e, 2, 64);
    e += __shfl_xor(e, 1, 64);
    return e;   // lane l: total of entry ((l>>5)&1)*8 + ((l>>4)&1)*4 + ((l>>3)&1)*2 + ((l>>2)&1)
}

__global__ __launch_bounds__(256) void enc_attn_kernel(const unsigned short* __restrict__ qkv, int S, int D, int heads,
                                                       float scale, unsigned short* __restrict__ out, int n_rows) {
    const int lane = threadIdx.x & 63;
    const int gw = (int)blockIdx.x * 4 + (threadIdx.x >> 6);       // one wavefront per (row, head)
    if (gw >= n_rows * heads) return;
    const int row = gw / heads, h = gw - row * heads;
    const int hd = D / heads, seq0 = (row / S) * S;
    const int ld = 3 * D;
    // lane owns the head columns {2 lane, 2 lane + 1} and {128 + 2 lane, 129 + 2 lane}: two 4-byte loads per row
    // (hd even, <= 256: every pair is valid or invalid as a whole)
    int col[2];
    float cm[2], q[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = 128 * u + 2 * lane;
        cm[u] = c < hd ? 1.f : 0.f;
        col[u] = c < hd ? c : 0;
        const unsigned pr = *reinterpret_cast<const unsigned*>(qkv + (int64_t)row * ld + h * hd + col[u]);
        q[2 * u] = bf2f((unsigned short)(pr & 0xffffu)) * cm[u] * scale;
        q[2 * u + 1] = bf2f((unsigned short)(pr >> 16)) * cm[u] * scale;
    }
    float mx = -INFINITY, den = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
    const int my_entry = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    for (int j0 = 0; j0 < S; j0 += 16) {
        unsigned kp[16][2], vp[16][2];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned short* kr = qkv + (int64_t)(seq0 + min(j0 + u, S - 1)) * ld + D + h * hd;
            kp[u][0] = *reinterpret_cast<const unsigned*>(kr + col[0]);
            kp[u][1] = *reinterpret_cast<const unsigned*>(kr + col[1]);
            vp[u][0] = *reinterpret_cast<const unsigned*>(kr + D + col[0]);
            vp[u][1] = *reinterpret_cast<const unsigned*>(kr + D + col[1]);
        }
        float part[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            part[u] = q[0] * __builtin_bit_cast(float, kp[u][0] << 16) + q[1] * __builtin_bit_cast(float, kp[u][0] & 0xffff0000u) +
                      q[2] * __builtin_bit_cast(float, kp[u][1] << 16) + q[3] * __builtin_bit_cast(float, kp[u][1] & 0xffff0000u);
        float sc = butterfly16_sum(part, lane);            // score of key j0 + my_entry
        if (j0 + my_entry >= S) sc = -INFINITY;
        const float bmx = wave_max(sc);
        const float nmx = fmaxf(mx, bmx);
        const float resc = expf(mx - nmx);
        den *= resc;
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] *= resc;
        mx = nmx;
        const float pw = expf(sc - mx);                    // 0 for masked entries
        den += wave_sum((lane & 3) == 0 ? pw : 0.f);       // every 4 lanes hold the same entry: one copy each
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int src = ((u >> 3) & 1) * 32 + ((u >> 2) & 1) * 16 + ((u >> 1) & 1) * 8 + (u & 1) * 4;
            const float pu = __shfl(pw, src, 64);
            o[0] += pu * __builtin_bit_cast(float, vp[u][0] << 16);
            o[1] += pu * __builtin_bit_cast(float, vp[u][0] & 0xffff0000u);
            o[2] += pu * __builtin_bit_cast(float, vp[u][1] << 16);
            o[3] += pu * __builtin_bit_cast(float, vp[u][1] & 0xffff0000u);
        }
    }
    const float inv = 1.0f / den;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (128 * u + 2 * lane < hd) {
            const unsigned pk = (unsigned)f2bf(o[2 * u] * inv) | ((unsigned)f2bf(o[2 * u + 1] * inv) << 16);
            *reinterpret_cast<unsigned*>(out + (int64_t)row * D + h * hd + 128 * u + 2 * lane) = pk;
        }
}

// Same, any head dimension <= 256 (odd ones such as 712 / 8 = 89): 2-byte loads, lane owns columns lane + 64 u.
__global__ __launch_bounds__(256) void enc_attn_generic_kernel(const unsigned short* __restrict__ qkv, int S, int D, int heads,
                                                               float scale, unsigned short* __restrict__ out, int n_rows) {
    const int lane = threadIdx.x & 63;
    const int gw = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= n_rows * heads) return;
    const int row = gw / heads, h = gw - row * heads;
    const int hd = D / heads, seq0 = (row / S) * S;
    const int ld = 3 * D;
    int col[4];
    float cm[4], q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        col[u] = min(lane + 64 * u, hd - 1);
        cm[u] = lane + 64 * u < hd ? 1.f : 0.f;
        q[u] = bf2f(qkv[(int64_t)row * ld + h * hd + col[u]]) * cm[u] * scale;
    }
    float mx = -INFINITY, den = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
    const int my_entry = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    for (int j0 = 0; j0 < S; j0 += 16) {
        float part[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned short* kr = qkv + (int64_t)(seq0 + min(j0 + u, S - 1)) * ld + D + h * hd;
            part[u] = q[0] * bf2f(kr[col[0]]) + q[1] * bf2f(kr[col[1]]) + q[2] * bf2f(kr[col[2]]) + q[3] * bf2f(kr[col[3]]);
        }
        float sc = butterfly16_sum(part, lane);
        if (j0 + my_entry >= S) sc = -INFINITY;
        const float nmx = fmaxf(mx, wave_max(sc));
        const float resc = expf(mx - nmx);
        den *= resc;
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] *= resc;
        mx = nmx;
        const float pw = expf(sc - mx);
        den += wave_sum((lane & 3) == 0 ? pw : 0.f);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int src = ((u >> 3) & 1) * 32 + ((u >> 2) & 1) * 16 + ((u >> 1) & 1) * 8 + (u & 1) * 4;
            const float pu = __shfl(pw, src, 64);
            const unsigned short* vr = qkv + (int64_t)(seq0 + min(j0 + u, S - 1)) * ld + 2 * D + h * hd;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] += pu * bf2f(vr[col[c]]) * cm[c];
        }
    }
    const float inv = 1.0f / den;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < hd) out[(int64_t)row * D + h * hd + lane + 64 * u] = f2bf(o[u] * inv);
}

// y = LayerNorm(a + b) * gamma + beta over rows of width D (one wavefront per row); fp32 and bf16 copies
__global__ __launch_bounds__(256) void enc_add_ln_kernel(const float* __restrict__ a, const float* __restrict__ b, int D,
                                                         int n_rows, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         float* __restrict__ yf, unsigned short* __restrict__ yh) {
    const int lane = threadIdx.x & 63, row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    constexpr int MAXC = 32;          // D <= 2048
    float v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
        const int c = lane + 64 * u;
        v[u] = c < D ? a[(int64_t)row * D + c] + b[(int64_t)row * D + c] : 0.f;
        s += v[u];
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
        const float d = lane + 64 * u < D ? v[u] - mean : 0.f;
        ss += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
        const int c = lane + 64 * u;
        if (c < D) {
            const float y = (v[u] - mean) * rstd * gamma[c] + beta[c];
            yf[(int64_t)row * D + c] = y;
            yh[(int64_t)row * D + c] = f2bf(y);
        }
    }
}

__global__ __launch_bounds__(256) void enc_to_bf16_kernel(const float* __restrict__ x, int64_t n, unsigned short* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = f2bf(x[i]);
}

}  // namespace

// the GEMM's 64 KB of dynamic LDS has to be allowed once per kernel instantiation
template <typename Kern>
static bool eg_allow_lds(Kern kern) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, EG_LDS_BYTES) == hipSuccess;
}

extern "C" int erc_enc_to_bf16(const float* x, int64_t n, void* y, void* stream) {
    ERC_REQUIRE(x && y && n > 0, "enc_to_bf16: bad arguments");
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(enc_to_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, (unsigned short*)y);
    ERC_LAUNCH_CHECK("enc_to_bf16");
    return ERC_OK;
}

static int eg_launch(const void* A, int lda, const void* W, int ldw, const float* bias, float* C_f32, void* C_bf16, int ldc, int M,
                     int N, int K, int relu, int epilogue, const EgEpi& ep, void* stream, const char* who) {
    ERC_REQUIRE(A && W && (C_f32 || C_bf16) && M > 0 && N > 0 && K >= 8, "%s: bad arguments (K >= 8)", who);
    ERC_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)A & 7) == 0 && ((uintptr_t)W & 7) == 0,
                "%s: K / pitches must be multiples of 4 elements, operands 8-byte aligned", who);
    static const bool ok0 = eg_allow_lds(enc_gemm_kernel<0>), ok1 = eg_allow_lds(enc_gemm_kernel<1>), ok2 = eg_allow_lds(enc_gemm_kernel<2>);
    ERC_REQUIRE(ok0 && ok1 && ok2, "%s: %d bytes of dynamic LDS refused", who, EG_LDS_BYTES);
    const dim3 grid(8 * erc_cdiv((int64_t)erc_cdiv(N, EG_BN) * erc_cdiv(M, EG_BM), 8));
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, dim3(512), EG_LDS_BYTES, (hipStream_t)stream, (const unsigned short*)A, lda,
                           (const unsigned short*)W, ldw, bias, C_f32, (unsigned short*)C_bf16, ldc, M, N, K, relu, ep);
    };
    if (epilogue == 1) launch(enc_gemm_kernel<1>);
    else if (epilogue == 2) launch(enc_gemm_kernel<2>);
    else launch(enc_gemm_kernel<0>);
    ERC_LAUNCH_CHECK(who);
    return ERC_OK;
}

extern "C" int erc_enc_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, float* C_f32, void* C_bf16,
                                 int ldc, int M, int N, int K, int relu, void* stream) {
    return eg_launch(A, lda, W, ldw, bias, C_f32, C_bf16, ldc, M, N, K, relu, 0, EgEpi{}, stream, "enc_gemm_bf16");
}

extern "C" int erc_enc_gemm_bf16_ex(const void* A, int lda, const void* W, int ldw, const float* bias, float* C_f32, void* C_bf16,
                                    int ldc, int M, int N, int K, int relu, int epilogue, const void* mask_src, int ld_mask,
                                    float scale, float drop_p, const uint64_t* rng_state, uint64_t rng_stream, void* stream) {
    ERC_REQUIRE(epilogue == 1 ? (rng_state && drop_p > 0.f && drop_p < 1.f) : epilogue == 2 ? (mask_src && ld_mask >= N) : epilogue == 0,
                "enc_gemm_bf16_ex: epilogue %d", epilogue);
    const EgEpi ep{(const unsigned short*)mask_src, ld_mask, scale, drop_p, rng_state, rng_stream};
    return eg_launch(A, lda, W, ldw, bias, C_f32, C_bf16, ldc, M, N, K, relu, epilogue, ep, stream, "enc_gemm_bf16_ex");
}

extern "C" int erc_enc_attention(const void* qkv, int n_seq, int S, int D, int heads, void* out, void* stream) {
    ERC_REQUIRE(qkv && out && n_seq > 0 && S > 0 && heads > 0 && D % heads == 0 && D / heads <= 256,
                "enc_attention: bad arguments (head dim <= 256)");
    const int n_rows = n_seq * S, hd = D / heads;
    const float scale = 1.0f / sqrtf((float)hd);
    const dim3 grid(erc_cdiv((int64_t)n_rows * heads, 4));
    const bool pairs = hd % 2 == 0 && D % 2 == 0 && ((uintptr_t)qkv & 3) == 0 && ((uintptr_t)out & 3) == 0;
    if (pairs)
        hipLaunchKernelGGL(enc_attn_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)qkv, S, D, heads,
                           scale, (unsigned short*)out, n_rows);
    else
        hipLaunchKernelGGL(enc_attn_generic_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)qkv, S, D,
                           heads, scale, (unsigned short*)out, n_rows);
    ERC_LAUNCH_CHECK("enc_attention");
    return ERC_OK;
}

extern "C" int erc_enc_add_layernorm(const float* a, const float* b, int D, int n_rows, const float* gamma, const float* beta,
                                     float eps, float* y_f32, void* y_bf16, void* stream) {
    ERC_REQUIRE(a && b && gamma && beta && y_f32 && y_bf16 && n_rows > 0 && D > 0 && D <= 2048, "enc_add_layernorm: bad arguments");
    hipLaunchKernelGGL(enc_add_ln_kernel, dim3(erc_cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, a, b, D, n_rows, gamma,
                       beta, eps, y_f32, (unsigned short*)y_bf16);
    ERC_LAUNCH_CHECK("enc_add_layernorm");
    return ERC_OK;
}
