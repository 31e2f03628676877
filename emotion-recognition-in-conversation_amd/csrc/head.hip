// K5: the classifier head of COGMEN as one row-tile kernel.
//
//   H3 = LeakyReLU(BatchNorm(H2))                      gcn.bn + leaky_relu      track_mm/cogmen.py:67-68,72-73
//   Z  = Dropout(ReLU(H3 W0^T + b0))                   cls[0..2]                track_mm/cogmen.py:116-121
//   logits = Z W3^T + b3 ; loss = cross_entropy        cls[3], F.cross_entropy  track_mm/cogmen.py:122,185
//   and the backward of all of it down to dY = dL/d(BatchNorm output), plus the two column sums BatchNorm's
//   backward needs (sum dY, sum dY*xhat) -- what loss.backward() does for these layers (cogmen.py:187-188).
//
// Given the batch statistics, every row is independent until BatchNorm's backward reduction, so a 16-row tile goes
// through the whole chain on chip: two 16x100x100 products on v_mfma_f32_16x16x4_f32 (W0 staged once per workgroup in
// LDS, read row-wise for the forward and column-wise for the backward), the 100 -> C product and its transpose on the
// VALU straight from the MFMA accumulator layout, one LDS transposition of the 16 x 100 dZ tile between the MFMA
// products.  The 7 column tiles of a row tile are spread over 4 wavefronts (a single wavefront per row tile measured
// 40 us: ~8000 instructions of serial VALU work); they meet in the 8-wide class space through LDS.  Replaces five launches (BN apply, Linear, Linear+CE, dgrad Linear,
// BN-backward statistics); the activations the weight-gradient launch needs (H3, Z, dZ, dlogits) are written once.
//
// Cross-workgroup sums (BatchNorm backward, loss, accuracy): per-workgroup partials written with write-through (sc1)
// stores, an arrival counter, and the last workgroup adds them in workgroup order in fp64 -- deterministic, no fences.
#include "erc_common.h"
#include <stdlib.h>

extern "C" int erc_head_fused_rows_per_workgroup(int n_rows);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HF_S = 108;      // LDS row pitch in floats: 16-byte aligned rows, 108 mod 64 = 44 -> conflict-free b128 rows
constexpr int HF_ST = 116;     // pitch of the dZ row tiles: >= 112 (7 full column tiles are written), 116 mod 64 = 52
constexpr int HF_NT = 7;       // column tiles of 16 (F <= 100 -> 7 tiles, the last one partial)
constexpr int HF_MAXF = 100;
constexpr int HF_MAXC = 8;
constexpr int HF_PART = 2 * 112 + 4;  // floats per workgroup partial record: column sums of dY | of dY * xhat | loss part, hits, weight sum, pad
constexpr int BS_G = 256;      // workgroups of the statistics pass: capacity of the partial records (one per CU for N > 8 192 rows)
constexpr int BS_G_SMALL = 64; // ... up to 8 192 rows (the last arriver's sum over the partials is the longer part there)

__device__ __forceinline__ float ld_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1d(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1d(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// Training-mode BatchNorm statistics: column mean / rstd of x [N,F] into saved[2F], running statistics updated
// (torch.nn.BatchNorm1d, cogmen.py:67).  One launch: fp64 partials per workgroup, finalised by the last arriver.
__global__ __launch_bounds__(256) void bn_batch_stats_kernel(const float* __restrict__ x, int ldx, int N, int F,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum, float eps,
                                                             float* __restrict__ saved, double* partial, int* counter) {
    __shared__ double sh[2][128];
    __shared__ int s_last;
    const int tid = threadIdx.x, c = tid & 127, half = tid >> 7;
    const int cc = min(c, F - 1);
    double a = 0.0, b = 0.0;
    // rows blockIdx*2 + half + 2G*i: eight unconditional (row-clamped, masked by multiplication) loads in flight
    const int stride = 2 * (int)gridDim.x;
    for (int row0 = (int)blockIdx.x * 2 + half; row0 < N; row0 += 8 * stride) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = x[(int64_t)min(row0 + j * stride, N - 1) * ldx + cc];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double t = (double)v[j] * (row0 + j * stride < N ? 1.0 : 0.0);
            a += t, b += t * t;
        }
    }
    if (half == 1) sh[0][c] = a, sh[1][c] = b;
    __syncthreads();
    if (half == 0 && c < F) {
        st_sc1d(partial + (int64_t)blockIdx.x * 2 * F + c, a + sh[0][c]);
        st_sc1d(partial + (int64_t)blockIdx.x * 2 * F + F + c, b + sh[1][c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (int)gridDim.x - 1;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // thread (c, half): half 0 sums x, half 1 sums x^2, over the workgroups in order, 8 loads in flight
    double s = 0.0;
    if (c < F) {
        const int G = (int)gridDim.x;
        for (int g0 = 0; g0 < G; g0 += 32) {   // (64 doubles in flight per thread measured slower: 9.7 vs 7.4 us)
            double t[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) t[j] = ld_sc1d(partial + (int64_t)min(g0 + j, G - 1) * 2 * F + half * F + c);
#pragma unroll
            for (int j = 0; j < 32; ++j) s += t[j] * (g0 + j < G ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    sh[half][c] = s;
    __syncthreads();
    if (half == 0 && c < F) {
        const double m = sh[0][c] / (double)N;
        double var = sh[1][c] / (double)N - m * m;
        if (var < 0.0) var = 0.0;
        saved[c] = (float)m;
        saved[F + c] = (float)(1.0 / sqrt(var + (double)eps));
        const double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct HeadP {
    const float* H2;
    const float* gamma;
    const float* beta;
    const float* saved;  // [0,F) mean, [F,2F) rstd of the batch
    const float* W0;
    const float* b0;
    const float* W3;
    const float* b3;
    const int64_t* labels;
    const float* weight;  // class weights or null
    const uint64_t* rng;  // {offset, seed}, read when drop_p > 0
    float* H3;
    float* Z;
    float* logits;
    float* dlogits;
    float* dZ;
    float* dY;
    float* part;
    int* counter;
    float* bn_bwd;  // [0,F) mean of dY, [F,2F) mean of dY*xhat
    float* dgamma;
    float* dbeta;
    float* stats;
    float slope, drop_p;
    int ldh, N, F, C;
    // BNP: the batch statistics are finalised here from per-tile column sums (erc_cogmen_fwd_tile, bn mode 2)
    const float* bn_part;    // [bn_tiles][2F]: sum x | sum x^2 of 16 rows each
    float* saved_out;        // [2F] mean | rstd, written by workgroup 0 (the backward reads it)
    float* running_mean;
    float* running_var;
    float momentum, eps;
    int bn_tiles;
    int defer;               // 1: leave the workgroup records for the consumer to add up (erc_cogmen_bwd_tile), no last arriver
    uint64_t* stamps;        // diagnostic phase stamps of the middle workgroup (tools/cogmen_stamps.py) or null
    // bf16 copies of the weight-gradient operands (COGMEN bf16 compute mode, csrc/wgrad_bf16.hip), or null: H3, Z, dZ
    // [N, ldb16 >= F] and dlogits [N, 8]; pad columns are left untouched
    unsigned short *H3b, *Zb, *dZb, *dlb;
    int ldb16;
    const int32_t* n_dev;        // capacity mode: the true row count lives on the device (N = the capacity the grid is sized for)
    const int32_t* label_rows;   // labels[label_rows[row]] instead of labels[row] (labels kept in a resident store), or null
    int lddl;                    // row pitch of dlogits (>= C; 8 in the split compute modes: 16-byte rows for the weight-gradient launch)
};

#define HF_STAMP(slot)                                                                                              \
    do {                                                                                                            \
        if (p.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) p.stamps[slot] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

__device__ __forceinline__ unsigned short hf_bf(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}

// DPP lane exchanges inside a row of 16 lanes (VALU, no LDS traffic)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_X1 = 0xB1, DPP_X2 = 0x4E, DPP_HMIR = 0x141, DPP_MIR = 0x140;  // quad_perm xor 1 / xor 2, row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {  // total over the 16 lanes r, in every lane
    v += dpp_f<DPP_X1>(v), v += dpp_f<DPP_X2>(v), v += dpp_f<DPP_HMIR>(v), v += dpp_f<DPP_MIR>(v);
    return v;
}
__device__ __forceinline__ float row8_sum(float v) {  // total over each half row of 8 lanes
    v += dpp_f<DPP_X1>(v), v += dpp_f<DPP_X2>(v), v += dpp_f<DPP_HMIR>(v);
    return v;
}
__device__ __forceinline__ float row8_max(float v) {
    v = fmaxf(v, dpp_f<DPP_X1>(v)), v = fmaxf(v, dpp_f<DPP_X2>(v)), v = fmaxf(v, dpp_f<DPP_HMIR>(v));
    return v;
}
__device__ __forceinline__ float row8_min(float v) {
    v = fminf(v, dpp_f<DPP_X1>(v)), v = fminf(v, dpp_f<DPP_X2>(v)), v = fminf(v, dpp_f<DPP_HMIR>(v));
    return v;
}

// Workgroup = 8 wavefronts = 32 rows: wavefront w works on row tile rt = w >> 2 (16 rows) and column tiles
// nt = 2 (w & 3), 2 (w & 3) + 1 of the 7; the class-space quantities (8 wide) are exchanged through LDS.
template <bool BNP, int RPW>
__global__ __launch_bounds__(512) void head_fused_kernel(const HeadP p) {
    constexpr int NCW = 8 / RPW;   // wavefronts per row tile; each takes RPW of the 7 column tiles
    __shared__ __attribute__((aligned(16))) float sW[HF_MAXF * HF_S];  // W0, row pitch HF_S
    __shared__ __attribute__((aligned(16))) float sT[RPW][16 * HF_ST > 2048 ? 16 * HF_ST : 2048];   // (>= 8 KB: the BatchNorm sums stage uses it first)    // dZ row tiles (transposition between the MFMA products); sT[0] first holds sLg
    __shared__ __attribute__((aligned(16))) float sD[RPW][16][8];  // dlogits of the two row tiles
    __shared__ float sCol[RPW][2][112];
    __shared__ float sLoss[RPW][2];
    __shared__ double sRed[8];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, rt = RPW == 2 ? w >> 2 : 0, cq = RPW == 2 ? w & 3 : w;
    const int r = lane & 15, g = lane >> 4;
    const int F = p.F, C = p.C, N = p.n_dev ? min(max(*p.n_dev, 1), p.N) : p.N;
    float* const sLg = &sT[0][0];  // [rt][cq][16 rows][8 classes] partial logits, dead before sT is written
    HF_STAMP(0);
    __shared__ __attribute__((aligned(16))) float sSaved[2 * HF_MAXF];
    __shared__ __attribute__((aligned(16))) float sGB[2 * HF_MAXF];      // BNP: gamma | beta, staged with the first loads

    // the H2 rows of the A fragments are requested first: they do not depend on the BatchNorm statistics computed below
    const int m0 = ((int)blockIdx.x * RPW + rt) * 16;
    const int mrow = m0 + r, mrc = min(mrow, N - 1);
    f32x4 xh[HF_NT];
#pragma unroll
    for (int kb = 0; kb < HF_NT; ++kb) {
        const int k0 = 16 * kb + 4 * g;
        xh[kb] = *reinterpret_cast<const f32x4*>(p.H2 + (int64_t)mrc * p.ldh + (k0 < F ? k0 : 0));
    }
    // ---- stage W0 (F x F) into LDS: F*F/4 <= 2500 float4 over 512 threads, all loads in flight
    {
        const int nq = F * F / 4, per_row = F / 4;
        f32x4 v[5];
        int dst[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = min(j * 512 + tid, nq - 1);
            const int row = i / per_row, q = i - row * per_row;
            v[j] = *reinterpret_cast<const f32x4*>(p.W0 + (int64_t)row * F + 4 * q);
            dst[j] = row * HF_S + 4 * q;
        }
        // ---- BNP: training-mode BatchNorm statistics (torch.nn.BatchNorm1d, cogmen.py:67) from the per-tile column
        //      sums the forward tile kernel left behind: every workgroup adds the tiles in order (fp64) -- no last
        //      arriver, no extra launch; workgroup 0 publishes mean | rstd and updates the running statistics.
        //      Thread (pair = tid % 128 < F, part = tid / 128) takes two columns and a quarter of the tile list: one
        //      batch of 8-byte loads, in flight together with the W0 loads above.
        if (BNP) {
            // gamma / beta -> LDS (visible behind the barriers below): the A fragments then need no global operand but H2
            if (tid < 2 * (F / 4)) {
                const int which = tid / (F / 4), q4 = tid - which * (F / 4);
                *reinterpret_cast<f32x4*>(sGB + which * F + 4 * q4) = *reinterpret_cast<const f32x4*>((which ? p.beta : p.gamma) + 4 * q4);
            }
            double* const sBn = reinterpret_cast<double*>(&sT[0][0]);   // [4][2][128] doubles = 8 KB of the tile area
            const int pair = tid & 127, part = tid >> 7;
            const int G = p.bn_tiles, Gq = (G + 3) >> 2;
            const int g_begin = min(part * Gq, G), g_end = min(G, g_begin + Gq);
            double ax = 0.0, ay = 0.0;
            if (pair < F) {
                for (int g0 = g_begin; g0 < g_end; g0 += 32) {
                    float2 t[32];
#pragma unroll
                    for (int j = 0; j < 32; ++j) t[j] = *reinterpret_cast<const float2*>(p.bn_part + (int64_t)min(g0 + j, G - 1) * 2 * F + 2 * pair);
#pragma unroll
                    for (int j = 0; j < 32; ++j) {
                        const double mk = g0 + j < g_end ? 1.0 : 0.0;
                        ax += (double)t[j].x * mk, ay += (double)t[j].y * mk;
                    }
                }
            }
            sBn[(part * 2 + 0) * 128 + pair] = ax, sBn[(part * 2 + 1) * 128 + pair] = ay;
            __syncthreads();
            if (tid < F) {
                // column c: sum x = slot c, sum x^2 = slot F + c of the 2F slots; slot s lives in pair s / 2, component s % 2
                auto slot_sum = [&](int sl) {
                    const int pr = sl >> 1, cm = sl & 1;
                    return ((sBn[(0 * 2 + cm) * 128 + pr] + sBn[(1 * 2 + cm) * 128 + pr]) + sBn[(2 * 2 + cm) * 128 + pr]) + sBn[(3 * 2 + cm) * 128 + pr];
                };
                const double sx = slot_sum(tid), sxx = slot_sum(F + tid);
                const double m = sx / (double)N;
                double var = sxx / (double)N - m * m;
                if (var < 0.0) var = 0.0;
                const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
                sSaved[tid] = mean, sSaved[F + tid] = rstd;
                if (blockIdx.x == 0) {
                    p.saved_out[tid] = mean, p.saved_out[F + tid] = rstd;
                    const double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
                    p.running_mean[tid] = (1.f - p.momentum) * p.running_mean[tid] + p.momentum * mean;
                    p.running_var[tid] = (1.f - p.momentum) * p.running_var[tid] + p.momentum * (float)unbiased;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (j * 512 + tid < nq) *reinterpret_cast<f32x4*>(sW + dst[j]) = v[j];
    }
    // ---- cross-entropy normaliser: sum of the class weights of all rows (every workgroup computes it)
    if (p.weight) {
        double wacc = 0.0;
        for (int i0 = 0; i0 < N; i0 += 4 * 512) {
            int y[4];
            float wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = min(i0 + j * 512 + tid, N - 1);
                y[j] = (int)p.labels[p.label_rows ? p.label_rows[rr] : rr];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = p.weight[y[j]];
#pragma unroll
            for (int j = 0; j < 4; ++j) wacc += (double)wv[j] * (i0 + j * 512 + tid < N ? 1.0 : 0.0);
        }
        wacc = wave_sum_d(wacc);
        if (lane == 0) sRed[w] = wacc;
    }

    // ---- P1: A fragments of H3 = lrelu(bn(H2)); lane (r,g): row m0 + r, k = 16 kb + 4 g + t
    float a1[HF_NT][4];
#pragma unroll
    for (int kb = 0; kb < HF_NT; ++kb) {
        const int k0 = 16 * kb + 4 * g;
        const bool kv = k0 < F;
        const int k0c = kv ? k0 : 0;
        const float km = kv ? 1.f : 0.f;
        const f32x4 x = xh[kb];
        const f32x4 mu = BNP ? *reinterpret_cast<const f32x4*>(sSaved + k0c) : *reinterpret_cast<const f32x4*>(p.saved + k0c);
        const f32x4 rs = BNP ? *reinterpret_cast<const f32x4*>(sSaved + F + k0c) : *reinterpret_cast<const f32x4*>(p.saved + F + k0c);
        const f32x4 ga = BNP ? *reinterpret_cast<const f32x4*>(sGB + k0c) : *reinterpret_cast<const f32x4*>(p.gamma + k0c);
        const f32x4 be = BNP ? *reinterpret_cast<const f32x4*>(sGB + F + k0c) : *reinterpret_cast<const f32x4*>(p.beta + k0c);
        f32x4 h;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float z = (x[t] - mu[t]) * rs[t] * ga[t] + be[t];
            h[t] = (z > 0.f ? z : z * p.slope);
            a1[kb][t] = h[t] * km;
        }
        if ((RPW == 2 ? (kb & 3) : kb) == cq && mrow < N && kv) {
            if (p.H3) *reinterpret_cast<f32x4*>(p.H3 + (int64_t)mrow * F + k0) = h;
            if (p.H3b) *reinterpret_cast<uint2*>(p.H3b + (int64_t)mrow * p.ldb16 + k0) =
                make_uint2((uint32_t)hf_bf(h[0]) | ((uint32_t)hf_bf(h[1]) << 16), (uint32_t)hf_bf(h[2]) | ((uint32_t)hf_bf(h[3]) << 16));
        }
    }
    // operands of the later phases, requested now so that their latency hides behind the first MFMA product
    int ylab[4];
    float wyq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = min(m0 + 4 * g + q, N - 1);
        ylab[q] = (int)p.labels[p.label_rows ? p.label_rows[rr] : rr];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) wyq[q] = p.weight ? p.weight[ylab[q]] : 1.f;
    int colc[RPW];
    float cm[RPW], x6[RPW][4], mu6[RPW], rs6[RPW], ga6[RPW], be6[RPW], b0c[RPW], w3[HF_MAXC][RPW];
#pragma unroll
    for (int jn = 0; jn < RPW; ++jn) {
        const int col = 16 * (RPW * cq + jn) + r;
        colc[jn] = min(col, F - 1);
        cm[jn] = col < F ? 1.f : 0.f;
        mu6[jn] = BNP ? sSaved[colc[jn]] : p.saved[colc[jn]], rs6[jn] = BNP ? sSaved[F + colc[jn]] : p.saved[F + colc[jn]];
        ga6[jn] = p.gamma[colc[jn]], be6[jn] = p.beta[colc[jn]];
        b0c[jn] = p.b0[colc[jn]];
#pragma unroll
        for (int q = 0; q < 4; ++q) x6[jn][q] = p.H2[(int64_t)min(m0 + 4 * g + q, N - 1) * p.ldh + colc[jn]];
#pragma unroll
        for (int c = 0; c < HF_MAXC; ++c) w3[c][jn] = p.W3[(int64_t)min(c, C - 1) * F + colc[jn]] * cm[jn] * (c < C ? 1.f : 0.f);
    }
    const int cr = r & 7;
    const float b3c = p.b3[min(cr, C - 1)];
    __syncthreads();  // sW, sRed complete
    HF_STAMP(1);
    double wsum = (double)N;
    if (p.weight) {
        wsum = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) wsum += sRed[j];
    }
    const float inv_w = (float)(1.0 / wsum);

    // ---- P2: Z = dropout(relu(H3 W0^T + b0)) for this wavefront's column tiles; rows m0 + 4g + q, columns 16 nt + r
    float zreg[RPW][4];
#pragma unroll
    for (int jn = 0; jn < RPW; ++jn)
#pragma unroll
        for (int q = 0; q < 4; ++q) zreg[jn][q] = 0.f;
    const float scale = 1.f / (1.f - p.drop_p);
    uint64_t rng_off = 0, rng_seed = 0;
    if (p.drop_p > 0.f) rng_off = p.rng[0], rng_seed = p.rng[1];
#pragma unroll
    for (int jn = 0; jn < RPW; ++jn) {
        const int nt = RPW * cq + jn;
        if (nt < HF_NT) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < HF_NT; ++kb) {
                const int k0 = 16 * kb + 4 * g;
                const int k0c = k0 < F ? k0 : 0;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(sW + colc[jn] * HF_S + k0c);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kb][t], wv[t], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + 4 * g + q;
                float z = fmaxf(acc[q] + b0c[jn], 0.f);
                if (p.drop_p > 0.f) {
                    const float u = erc_uniform(rng_seed, rng_off, (uint64_t)row * (uint64_t)F + (uint64_t)(16 * nt + r));
                    z = (u >= p.drop_p) ? z * scale : 0.f;
                }
                z *= cm[jn];
                zreg[jn][q] = z;
                if (row < N && 16 * nt + r < F) {
                    if (p.Z) p.Z[(int64_t)row * F + 16 * nt + r] = z;
                    if (p.Zb) p.Zb[(int64_t)row * p.ldb16 + 16 * nt + r] = hf_bf(z);
                }
            }
        }
    }

    // ---- P3a: this wavefront's share of the logits: dot products over its columns, summed over the 16 lanes r
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float mine = 0.f;
#pragma unroll
        for (int c = 0; c < HF_MAXC; ++c) {
            float zw = zreg[0][q] * w3[c][0];
            if (RPW == 2) zw += zreg[RPW - 1][q] * w3[c][RPW - 1];
            const float s = row16_sum(zw);
            if (c == cr) mine = s;
        }
        if (r < 8) sLg[((rt * NCW + cq) * 16 + 4 * g + q) * 8 + cr] = mine;
    }
    __syncthreads();
    HF_STAMP(2);
    // ---- P3b: cross entropy with one class per lane (lanes r and r + 8 duplicate); rows 4g + q
    float lsum = 0.f, hsum = 0.f;
    const bool cv = cr < C;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = m0 + 4 * g + q;
        const bool rv = row < N;
        float v = b3c;
#pragma unroll
        for (int j = 0; j < NCW; ++j) v += sLg[((rt * NCW + j) * 16 + 4 * g + q) * 8 + cr];
        const float mx = row8_max(cv ? v : -3.0e38f);
        const float se = row8_sum(cv ? expf(v - mx) : 0.f);
        const float lse = mx + logf(se);
        const int y = ylab[q];
        const float coef = rv ? wyq[q] * inv_w : 0.f;
        const float dq = cv ? coef * (expf(v - lse) - (cr == y ? 1.f : 0.f)) : 0.f;
        const float ly = row8_sum(cr == y ? v : 0.f);
        const float am = row8_min((cv && v == mx) ? (float)cr : 99.f);
        if (cq == 0) {
            if (r < 8) sD[rt][4 * g + q][cr] = dq;
            if (rv && r < C) {
                p.logits[(int64_t)row * C + r] = v;
                if (p.dlogits) p.dlogits[(int64_t)row * p.lddl + r] = dq;
                if (p.dlb) p.dlb[(int64_t)row * 8 + r] = hf_bf(dq);
            }
            if (rv && r == 0) lsum += wyq[q] * (lse - ly), hsum += ((int)am == y) ? 1.f : 0.f;
        }
    }
    if (cq == 0) {
        lsum = wave_sum(lsum), hsum = wave_sum(hsum);
        if (lane == 0) sLoss[rt][0] = lsum, sLoss[rt][1] = hsum;
    }
    __syncthreads();
    HF_STAMP(3);

    // ---- P4: dZ = (dlogits W3) * relu/dropout mask for this wavefront's columns; into the row tile in LDS
    float* const tile = sT[rt];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 dlo = *reinterpret_cast<const f32x4*>(&sD[rt][4 * g + q][0]);
        const f32x4 dhi = *reinterpret_cast<const f32x4*>(&sD[rt][4 * g + q][4]);
        const int row = m0 + 4 * g + q;
#pragma unroll
        for (int jn = 0; jn < RPW; ++jn) {
            const int nt = RPW * cq + jn;
            if (nt < HF_NT) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) s += dlo[c] * w3[c][jn];
#pragma unroll
                for (int c = 0; c < 4; ++c) s += dhi[c] * w3[4 + c][jn];
                const float dz = zreg[jn][q] > 0.f ? s * scale : 0.f;
                if (row < N && 16 * nt + r < F) {
                    if (p.dZ) p.dZ[(int64_t)row * F + 16 * nt + r] = dz;
                    if (p.dZb) p.dZb[(int64_t)row * p.ldb16 + 16 * nt + r] = hf_bf(dz);
                }
                tile[(4 * g + q) * HF_ST + 16 * nt + r] = dz;
            }
        }
    }
    __syncthreads();
    HF_STAMP(4);

    // ---- P5: dH3 = dZ W0 : A fragments from the transposed tile, B[k = j][n = i] = W0[j][i] read column-wise from LDS
    float a2[HF_NT][4];
#pragma unroll
    for (int kb = 0; kb < HF_NT; ++kb) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * HF_ST + 16 * kb + 4 * g);  // columns >= F hold zeros
#pragma unroll
        for (int t = 0; t < 4; ++t) a2[kb][t] = v[t];
    }
#pragma unroll
    for (int jn = 0; jn < RPW; ++jn) {
        const int nt = RPW * cq + jn;
        if (nt < HF_NT) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < HF_NT; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int jc = min(16 * kb + 4 * g + t, F - 1);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[kb][t], sW[jc * HF_S + colc[jn]], acc, 0, 0, 0);
                }
            // ---- P6: dY = dH3 * lrelu'(bn output), BatchNorm-backward column partials
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + 4 * g + q;
                const bool ok = row < N && 16 * nt + r < F;
                const float xh = (x6[jn][q] - mu6[jn]) * rs6[jn];
                const float zz = xh * ga6[jn] + be6[jn];
                const float dy = ok ? acc[q] * (zz > 0.f ? 1.f : p.slope) : 0.f;
                if (ok) p.dY[(int64_t)row * F + 16 * nt + r] = dy;
                a += dy, b += dy * xh;
            }
            a += __shfl_xor(a, 16, 64), a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64), b += __shfl_xor(b, 32, 64);
            if (g == 0) sCol[rt][0][16 * nt + r] = a, sCol[rt][1][16 * nt + r] = b;
        }
    }
    __syncthreads();
    HF_STAMP(5);
    float* const rec = p.part + (int64_t)blockIdx.x * HF_PART;
    if (tid < 112) {
        st_sc1(rec + tid, sCol[0][0][tid] + (RPW == 2 ? sCol[RPW - 1][0][tid] : 0.f));
        st_sc1(rec + 112 + tid, sCol[0][1][tid] + (RPW == 2 ? sCol[RPW - 1][1][tid] : 0.f));
    } else if (tid < 114) {
        st_sc1(rec + 224 + (tid - 112), sLoss[0][tid - 112] + (RPW == 2 ? sLoss[RPW - 1][tid - 112] : 0.f));
    } else if (tid == 114) {
        st_sc1(rec + 226, (float)wsum);
    }
    if (p.defer) return;     // (uniform) the records become visible with the end of the kernel; the next kernel adds them up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    HF_STAMP(6);
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(p.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (int)gridDim.x - 1;
        if (last) __hip_atomic_store(p.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- last arriver: column sums over the workgroups, in order, fp64.  thread c < F: both sums of column c;
    //      threads F, F+1: loss and accuracy
    const int G = (int)gridDim.x;
    const bool col = tid < F, aux = tid >= F && tid < F + 2;
    const int o1 = col ? tid : 224 + (tid - F), o2 = col ? 112 + tid : 224 + (tid - F);
    double s1 = 0.0, s2 = 0.0;
    if (col || aux) {
        for (int g0 = 0; g0 < G; g0 += 64) {   // 62 workgroups at B = 32: one batch of loads
            float t1[64], t2[64];
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const float* q = p.part + (int64_t)min(g0 + j, G - 1) * HF_PART;
                t1[j] = ld_sc1(q + o1), t2[j] = ld_sc1(q + o2);
            }
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const double m = g0 + j < G ? 1.0 : 0.0;
                s1 += (double)t1[j] * m, s2 += (double)t2[j] * m;
            }
        }
    }
    if (col) {
        p.bn_bwd[tid] = (float)(s1 / (double)N);
        p.bn_bwd[F + tid] = (float)(s2 / (double)N);
        p.dbeta[tid] = (float)s1;
        p.dgamma[tid] = (float)s2;
    } else if (tid == F) {
        p.stats[0] = (float)(s1 / wsum);
        p.stats[2] = (float)wsum;
    } else if (tid == F + 1) {
        p.stats[1] = (float)s1;
    }
    if (p.stamps && tid == 0) p.stamps[7] = __builtin_amdgcn_s_memrealtime();   // the last arriver, whichever workgroup it is
}

// ---------------------------------------------------------------------------------------------------------------
// THROUGHPUT FORM (more than 8 192 rows: B = 512 batches).  The kernel above spreads a 16-row tile over 4-7 wavefronts that meet
// four times in LDS: right when ONE round of workgroups is all there is (2 000 rows: the chain per wavefront sets the time),
// wrong when there are 8 rounds -- 125 us at 33 k rows against 13 us of fp32 matrix-core time, one workgroup per CU (147
// registers x 8 wavefronts; squeezed to 128 for two per CU it spilled: 226 us).  Here ONE WAVEFRONT takes a row tile through the
// whole chain (its 7 column tiles in turn; the 16 x 100 dZ tile is transposed through a wavefront-private LDS area: no
// workgroup barrier inside the loop), wavefronts loop over tiles, W0 is staged once per workgroup, and two workgroups of four
// wavefronts share a CU: the dependent chain of one tile hides behind the other wavefronts' tiles.  Same arithmetic, same
// outputs and records as head_fused_kernel<false, .> (the per-row math is copied instruction for instruction).
constexpr int HR_ST = 112;     // pitch of a wavefront's dZ tile (>= 112: 7 full column tiles are written)
__global__ __launch_bounds__(256, 2) void head_rows_kernel(const HeadP p) {
    __shared__ __attribute__((aligned(16))) float sW[HF_MAXF * HF_S];      // W0, row pitch HF_S
    __shared__ __attribute__((aligned(16))) float sT[4][16 * HR_ST];       // per wavefront: the dZ tile (transposition between the products)
    __shared__ __attribute__((aligned(16))) float sD[4][16][8];            // per wavefront: dlogits of its tile
    float (*const sCol)[2][112] = reinterpret_cast<float (*)[2][112]>(&sT[0][0]);   // [4][2][112] column sums: in the tile area, after the loop
    __shared__ float sLoss[4][2];                                          //   (79 240 bytes in all: two workgroups per CU -- at 81 288 the second one was not admitted)
    __shared__ double sRed[4];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int F = p.F, C = p.C, N = p.n_dev ? min(max(*p.n_dev, 1), p.N) : p.N;
    // diagnostic phase stamps (tools/head_rows_stamps.py): wavefront 0 of the middle workgroup, its first tile
#define HR_STAMP(slot)                                                                                            \
    do {                                                                                                          \
        if (p.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) p.stamps[slot] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    HR_STAMP(0);
    {   // W0 (F x F) -> LDS
        const int nq = F * F / 4, per_row = F / 4;
        for (int i0 = 0; i0 < nq; i0 += 5 * 256) {
            f32x4 v[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int i = min(i0 + j * 256 + tid, nq - 1);
                const int row = i / per_row, q = i - row * per_row;
                v[j] = *reinterpret_cast<const f32x4*>(p.W0 + (int64_t)row * F + 4 * q);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int i = i0 + j * 256 + tid;
                const int row = min(i, nq - 1) / per_row, q = min(i, nq - 1) - row * per_row;
                if (i < nq) *reinterpret_cast<f32x4*>(sW + row * HF_S + 4 * q) = v[j];
            }
        }
    }
    if (p.weight) {   // cross-entropy normaliser: sum of the class weights of all rows (every workgroup computes it)
        double wacc = 0.0;
        for (int i0 = 0; i0 < N; i0 += 4 * 256) {
            int y[4];
            float wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = min(i0 + j * 256 + tid, N - 1);
                y[j] = (int)p.labels[p.label_rows ? p.label_rows[rr] : rr];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = p.weight[y[j]];
#pragma unroll
            for (int j = 0; j < 4; ++j) wacc += (double)wv[j] * (i0 + j * 256 + tid < N ? 1.0 : 0.0);
        }
        wacc = wave_sum_d(wacc);
        if (lane == 0) sRed[w] = wacc;
    }
    // per-column constants (column 16 nt + r of every column tile) live in LDS and are read where they are used: held in
    // registers across the tile loop (105 of them) the kernel spilled 210
    __shared__ __attribute__((aligned(16))) float sPar[6][112];      // mean | rstd | gamma | beta | b0 | column mask, columns >= F clamped (mask 0)
    __shared__ float sW3[HF_MAXC][112];
    if (tid < 112) {
        const int cc = min(tid, F - 1);
        sPar[0][tid] = p.saved[cc], sPar[1][tid] = p.saved[F + cc], sPar[2][tid] = p.gamma[cc], sPar[3][tid] = p.beta[cc];
        sPar[4][tid] = p.b0[cc], sPar[5][tid] = tid < F ? 1.f : 0.f;
#pragma unroll
        for (int c = 0; c < HF_MAXC; ++c) sW3[c][tid] = (tid < F && c < C) ? p.W3[(int64_t)c * F + cc] : 0.f;
    }
    const int cr = r & 7;
    const float b3c = p.b3[min(cr, C - 1)];
    const bool cv = cr < C;
    __syncthreads();  // sW, sRed complete
    double wsum = (double)N;
    if (p.weight) wsum = ((sRed[0] + sRed[1]) + sRed[2]) + sRed[3];
    const float inv_w = (float)(1.0 / wsum);
    const float scale = 1.f / (1.f - p.drop_p);
    uint64_t rng_off = 0, rng_seed = 0;
    if (p.drop_p > 0.f) rng_off = p.rng[0], rng_seed = p.rng[1];
    float* const tile = sT[w];
    float colA[HF_NT], colB[HF_NT];
#pragma unroll
    for (int nt = 0; nt < HF_NT; ++nt) colA[nt] = colB[nt] = 0.f;
    float lsum = 0.f, hsum = 0.f;
    const int n_tiles = (N + 15) / 16;
    HR_STAMP(1);
#pragma unroll 1
    for (int t_i = (int)blockIdx.x * 4 + w; t_i < n_tiles; t_i += (int)gridDim.x * 4) {
        // (the lane coordinates go through an opaque zero once per tile: otherwise the ~300 LDS / global offsets below, all
        //  invariant across tiles, are hoisted out of this loop and spilled -- 47 registers in scratch, reloaded inside the chain)
        int opaque0 = 0;
        asm volatile("" : "+v"(opaque0));
        const int r = (lane & 15) + opaque0, g = (lane >> 4) + opaque0;
        const int m0 = t_i * 16;
        const int mrow = m0 + r, mrc = min(mrow, N - 1);
        // ---- P1: A fragments of H3 = lrelu(bn(H2)); lane (r,g): row m0 + r, k = 16 kb + 4 g + t
        float a1[HF_NT][4];
        f32x4 xr[HF_NT];      // the tile's H2 rows: the only global operands of this phase (the BatchNorm constants come from LDS)
#pragma unroll
        for (int kb = 0; kb < HF_NT; ++kb) {
            const int k0 = 16 * kb + 4 * g;
            xr[kb] = *reinterpret_cast<const f32x4*>(p.H2 + (int64_t)mrc * p.ldh + (k0 < F ? k0 : 0));
        }
#pragma unroll
        for (int kb = 0; kb < HF_NT; ++kb) {
            const int k0 = 16 * kb + 4 * g;
            const bool kv = k0 < F;
            const int k0c = kv ? k0 : 0;
            const float km = kv ? 1.f : 0.f;
            const f32x4 x = xr[kb];
            const f32x4 mu = *reinterpret_cast<const f32x4*>(&sPar[0][k0c]), rs = *reinterpret_cast<const f32x4*>(&sPar[1][k0c]);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(&sPar[2][k0c]), be = *reinterpret_cast<const f32x4*>(&sPar[3][k0c]);
            f32x4 h;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float z = (x[t] - mu[t]) * rs[t] * ga[t] + be[t];
                h[t] = (z > 0.f ? z : z * p.slope);
                a1[kb][t] = h[t] * km;
            }
            if (mrow < N && kv) {
                if (p.H3) *reinterpret_cast<f32x4*>(p.H3 + (int64_t)mrow * F + k0) = h;
                if (p.H3b) *reinterpret_cast<uint2*>(p.H3b + (int64_t)mrow * p.ldb16 + k0) =
                    make_uint2((uint32_t)hf_bf(h[0]) | ((uint32_t)hf_bf(h[1]) << 16), (uint32_t)hf_bf(h[2]) | ((uint32_t)hf_bf(h[3]) << 16));
            }
        }
        int ylab[4];
        float wyq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = min(m0 + 4 * g + q, N - 1);
            ylab[q] = (int)p.labels[p.label_rows ? p.label_rows[rr] : rr];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) wyq[q] = p.weight ? p.weight[ylab[q]] : 1.f;
        HR_STAMP(2);
        // ---- P2: Z = dropout(relu(H3 W0^T + b0)), all 7 column tiles; rows m0 + 4g + q, columns 16 nt + r
        float zreg[HF_NT][4];
#pragma unroll
        for (int nt = 0; nt < HF_NT; ++nt) {
            __builtin_amdgcn_sched_barrier(0);      // (one column tile at a time: hoisting the next tiles' 49 weight reads spills)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < HF_NT; ++kb) {
                const int k0 = 16 * kb + 4 * g;
                const int k0c = k0 < F ? k0 : 0;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(sW + min(16 * nt + r, F - 1) * HF_S + k0c);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kb][t], wv[t], acc, 0, 0, 0);
            }
            const float b0v = sPar[4][16 * nt + r], cmv = sPar[5][16 * nt + r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + 4 * g + q;
                float z = fmaxf(acc[q] + b0v, 0.f);
                if (p.drop_p > 0.f) {
                    const float u = erc_uniform(rng_seed, rng_off, (uint64_t)row * (uint64_t)F + (uint64_t)(16 * nt + r));
                    z = (u >= p.drop_p) ? z * scale : 0.f;
                }
                z *= cmv;
                zreg[nt][q] = z;
                tile[(4 * g + q) * HR_ST + 16 * nt + r] = z;      // (accumulator layout -> row layout through the wavefront's tile)
            }
        }
        // Z out, row-wise: lane (r, g) holds row m0 + r, columns 16 kb + 4 g .. + 3 -- one 16-byte (fp32) / 8-byte (bf16) store per
        // 16 columns instead of four 4- / 2-byte stores in the accumulator layout (91 narrow stores per lane and tile were what
        // the products waited behind: vmcnt retires in order)
#pragma unroll
        for (int kb = 0; kb < HF_NT; ++kb) {
            const int k0 = 16 * kb + 4 * g;
            const f32x4 zv = *reinterpret_cast<const f32x4*>(tile + r * HR_ST + k0);
            if (mrow < N && k0 < F) {
                if (p.Z) *reinterpret_cast<f32x4*>(p.Z + (int64_t)mrow * F + k0) = zv;
                if (p.Zb) *reinterpret_cast<uint2*>(p.Zb + (int64_t)mrow * p.ldb16 + k0) =
                    make_uint2((uint32_t)hf_bf(zv[0]) | ((uint32_t)hf_bf(zv[1]) << 16), (uint32_t)hf_bf(zv[2]) | ((uint32_t)hf_bf(zv[3]) << 16));
            }
        }
        HR_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        // ---- P3: logits (class cr in lanes r and r + 8) and the cross entropy of rows 4g + q.  W3's columns come from LDS a class at
        //      a time (7 registers instead of 56 across P3 / P4)
        float mine4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < HF_MAXC; ++c) {
            float w3c[HF_NT];
#pragma unroll
            for (int nt = 0; nt < HF_NT; ++nt) w3c[nt] = sW3[c][16 * nt + r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zw = 0.f;
#pragma unroll
                for (int nt = 0; nt < HF_NT; ++nt) zw += zreg[nt][q] * w3c[nt];
                const float s_ = row16_sum(zw);
                if (c == cr) mine4[q] = s_;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = m0 + 4 * g + q;
            const bool rv = row < N;
            const float v = b3c + mine4[q];
            const float mx = row8_max(cv ? v : -3.0e38f);
            const float se = row8_sum(cv ? expf(v - mx) : 0.f);
            const float lse = mx + logf(se);
            const int y = ylab[q];
            const float coef = rv ? wyq[q] * inv_w : 0.f;
            const float dq = cv ? coef * (expf(v - lse) - (cr == y ? 1.f : 0.f)) : 0.f;
            const float ly = row8_sum(cr == y ? v : 0.f);
            const float am = row8_min((cv && v == mx) ? (float)cr : 99.f);
            if (r < 8) sD[w][4 * g + q][cr] = dq;
            if (rv && r < C) {
                p.logits[(int64_t)row * C + r] = v;
                if (p.dlogits) p.dlogits[(int64_t)row * p.lddl + r] = dq;
                if (p.dlb) p.dlb[(int64_t)row * 8 + r] = hf_bf(dq);
            }
            if (rv && r == 0) lsum += wyq[q] * (lse - ly), hsum += ((int)am == y) ? 1.f : 0.f;
        }
        HR_STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
        // ---- P4: dZ = (dlogits W3) * relu / dropout mask -> the wavefront's tile in LDS (written and read by this wavefront only)
        f32x4 dlo[4], dhi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) dlo[q] = *reinterpret_cast<const f32x4*>(&sD[w][4 * g + q][0]), dhi[q] = *reinterpret_cast<const f32x4*>(&sD[w][4 * g + q][4]);
#pragma unroll
        for (int nt = 0; nt < HF_NT; ++nt) {
            float w3n[HF_MAXC];
#pragma unroll
            for (int c = 0; c < HF_MAXC; ++c) w3n[c] = sW3[c][16 * nt + r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + 4 * g + q;
                float s_ = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) s_ += dlo[q][c] * w3n[c];
#pragma unroll
                for (int c = 0; c < 4; ++c) s_ += dhi[q][c] * w3n[4 + c];
                const float dz = zreg[nt][q] > 0.f ? s_ * scale : 0.f;
                tile[(4 * g + q) * HR_ST + 16 * nt + r] = dz;
            }
        }
        HR_STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        // ---- P5: dH3 = dZ W0 : A fragments from the transposed tile, B[k = j][n = i] = W0[j][i] read column-wise from LDS
        float a2[HF_NT][4];
#pragma unroll
        for (int kb = 0; kb < HF_NT; ++kb) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * HR_ST + 16 * kb + 4 * g);  // columns >= F hold zeros
#pragma unroll
            for (int t = 0; t < 4; ++t) a2[kb][t] = v[t];
            const int k0 = 16 * kb + 4 * g;      // dZ out, row-wise (as Z above)
            if (mrow < N && k0 < F) {
                if (p.dZ) *reinterpret_cast<f32x4*>(p.dZ + (int64_t)mrow * F + k0) = v;
                if (p.dZb) *reinterpret_cast<uint2*>(p.dZb + (int64_t)mrow * p.ldb16 + k0) =
                    make_uint2((uint32_t)hf_bf(v[0]) | ((uint32_t)hf_bf(v[1]) << 16), (uint32_t)hf_bf(v[2]) | ((uint32_t)hf_bf(v[3]) << 16));
            }
        }
#pragma unroll
        for (int nt = 0; nt < HF_NT; ++nt) {
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < HF_NT; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int jc = min(16 * kb + 4 * g + t, F - 1);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[kb][t], sW[jc * HF_S + min(16 * nt + r, F - 1)], acc, 0, 0, 0);
                }
            // ---- P6: dY = dH3 * lrelu'(bn output), BatchNorm-backward column partials of this wavefront
            float a = 0.f, b = 0.f;
            const float mu6 = sPar[0][16 * nt + r], rs6 = sPar[1][16 * nt + r], ga6 = sPar[2][16 * nt + r], be6 = sPar[3][16 * nt + r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + 4 * g + q;
                const bool ok = row < N && 16 * nt + r < F;
                const float x6 = p.H2[(int64_t)min(row, N - 1) * p.ldh + min(16 * nt + r, F - 1)];
                const float xh = (x6 - mu6) * rs6;
                const float zz = xh * ga6 + be6;
                const float dy = ok ? acc[q] * (zz > 0.f ? 1.f : p.slope) : 0.f;
                tile[(4 * g + q) * HR_ST + 16 * nt + r] = dy;      // (the tile is free: a2 was read in front of the products)
                a += dy, b += dy * xh;
            }
            colA[nt] += a, colB[nt] += b;
        }
        // dY out, row-wise
#pragma unroll
        for (int kb = 0; kb < HF_NT; ++kb) {
            const int k0 = 16 * kb + 4 * g;
            if (mrow < N && k0 < F) *reinterpret_cast<f32x4*>(p.dY + (int64_t)mrow * F + k0) = *reinterpret_cast<const f32x4*>(tile + r * HR_ST + k0);
        }
    }
    HR_STAMP(6);
    // ---- workgroup record: column sums over the wavefronts' tiles (lane groups g, then wavefronts), loss, hits
    __syncthreads();      // every wavefront is done with its tile area
    HR_STAMP(7);
#pragma unroll
    for (int nt = 0; nt < HF_NT; ++nt) {
        float a = colA[nt], b = colB[nt];
        a += __shfl_xor(a, 16, 64), a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64), b += __shfl_xor(b, 32, 64);
        if (g == 0) sCol[w][0][16 * nt + r] = a, sCol[w][1][16 * nt + r] = b;
    }
    lsum = wave_sum(lsum), hsum = wave_sum(hsum);
    if (lane == 0) sLoss[w][0] = lsum, sLoss[w][1] = hsum;
    __syncthreads();
    float* const rec = p.part + (int64_t)blockIdx.x * HF_PART;
    if (tid < 112) {
        st_sc1(rec + tid, ((sCol[0][0][tid] + sCol[1][0][tid]) + sCol[2][0][tid]) + sCol[3][0][tid]);
        st_sc1(rec + 112 + tid, ((sCol[0][1][tid] + sCol[1][1][tid]) + sCol[2][1][tid]) + sCol[3][1][tid]);
    } else if (tid < 114) {
        st_sc1(rec + 224 + (tid - 112), ((sLoss[0][tid - 112] + sLoss[1][tid - 112]) + sLoss[2][tid - 112]) + sLoss[3][tid - 112]);
    } else if (tid == 114) {
        st_sc1(rec + 226, (float)wsum);
    }
    if (p.defer) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    HR_STAMP(8);
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(p.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (int)gridDim.x - 1;
        if (last) __hip_atomic_store(p.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // last arriver: column sums over the workgroups, in order, fp64 (as head_fused_kernel)
    const int G = (int)gridDim.x;
    const bool col = tid < F, aux = tid >= F && tid < F + 2;
    const int o1 = col ? tid : 224 + (tid - F), o2 = col ? 112 + tid : 224 + (tid - F);
    double s1 = 0.0, s2 = 0.0;
    if (col || aux) {
        for (int g0 = 0; g0 < G; g0 += 32) {
            float t1[32], t2[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float* q = p.part + (int64_t)min(g0 + j, G - 1) * HF_PART;
                t1[j] = ld_sc1(q + o1), t2[j] = ld_sc1(q + o2);
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const double m = g0 + j < G ? 1.0 : 0.0;
                s1 += (double)t1[j] * m, s2 += (double)t2[j] * m;
            }
        }
    }
    if (col) {
        p.bn_bwd[tid] = (float)(s1 / (double)N);
        p.bn_bwd[F + tid] = (float)(s2 / (double)N);
        p.dbeta[tid] = (float)s1;
        p.dgamma[tid] = (float)s2;
    } else if (tid == F) {
        p.stats[0] = (float)(s1 / wsum);
        p.stats[2] = (float)wsum;
    } else if (tid == F + 1) {
        p.stats[1] = (float)s1;
    }
}

#undef HR_STAMP
// dx = gamma * rstd * (dY - mean(dY) - xhat * mean(dY * xhat)): the elementwise part of BatchNorm's backward
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, int ldx, int N, int F,
                                                           const float* __restrict__ gamma, const float* __restrict__ saved,
                                                           const float* __restrict__ bn_bwd, const float* __restrict__ dY,
                                                           int lddy, float* __restrict__ dx, int lddx) {
    const int per_row = F / 4;
    const int64_t total = (int64_t)N * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / per_row), c = 4 * (int)(i - (int64_t)row * per_row);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (int64_t)row * ldx + c);
        const f32x4 dy = *reinterpret_cast<const f32x4*>(dY + (int64_t)row * lddy + c);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(saved + c), rs = *reinterpret_cast<const f32x4*>(saved + F + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 ma = *reinterpret_cast<const f32x4*>(bn_bwd + c), mb = *reinterpret_cast<const f32x4*>(bn_bwd + F + c);
        f32x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float xh = (xv[t] - mu[t]) * rs[t];
            o[t] = ga[t] * rs[t] * (dy[t] - ma[t] - xh * mb[t]);
        }
        *reinterpret_cast<f32x4*>(dx + (int64_t)row * lddx + c) = o;
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int64_t erc_bn_batch_stats_ws_floats(int F) { return (int64_t)BS_G * 2 * F * 2 + 16; }

extern "C" int erc_bn_batch_stats(const float* x, int ldx, int N, int F, float* running_mean, float* running_var,
                                  float momentum, float eps, float* saved, float* ws, void* stream) {
    ERC_REQUIRE(x && running_mean && running_var && saved && ws, "bn_batch_stats: null pointer");
    ERC_REQUIRE(N > 0 && F > 0 && F <= 128 && ldx >= F, "bn_batch_stats: N=%d F=%d ldx=%d", N, F, ldx);
    ERC_REQUIRE(((uintptr_t)ws & 7) == 0, "bn_batch_stats: ws must be 8-byte aligned");
    const int cap = N > 8192 ? BS_G : BS_G_SMALL;      // B = 512 batches: 23 us on 64 CUs (4-byte loads), 13 MB to read
    const int grid = erc_cdiv(N, 8) < cap ? erc_cdiv(N, 8) : cap;
    int* counter = reinterpret_cast<int*>(ws + (int64_t)BS_G * 2 * F * 2);
    hipLaunchKernelGGL(bn_batch_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, N, F, running_mean,
                       running_var, momentum, eps, saved, reinterpret_cast<double*>(ws), counter);
    ERC_LAUNCH_CHECK("bn_batch_stats");
    return ERC_OK;
}

// Rows per workgroup of the fused head.  16 while the launch is latency-bound (n_rows <= 8192: every one of the 7 column
// tiles of a row tile has its own wavefront -- the per-wavefront chain is what sets the time there, 16.5 -> 13.1 us of work
// at 2 000 rows); 32 beyond (two row tiles, two column tiles per wavefront: half the workgroups, W0 staged half as often --
// 186 -> ~130 us at 33 000 rows).  ERC_HEAD_ROWS=16 / 32 forces one.
extern "C" int erc_head_fused_rows_per_workgroup(int n_rows) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("ERC_HEAD_ROWS");
        forced = e ? atoi(e) : 0;
    }
    if (forced == 16 || forced == 32) return forced;
    return n_rows <= 8192 ? 16 : 32;
}
extern "C" int64_t erc_head_fused_ws_floats(int n_rows) { return (int64_t)erc_cdiv(n_rows, 16) * HF_PART + 16; }

static uint64_t* g_head_stamps = nullptr;
// diagnostic: 8 x uint64 phase stamps (10 ns ticks) of the following erc_head_fused launches; NULL switches them off
extern "C" int erc_head_set_stamps(uint64_t* stamps) {
    g_head_stamps = stamps;
    return ERC_OK;
}

static int head_fused_launch(const float* H2, int ldh, int n_rows, int F, int C, const float* gamma, const float* beta,
                             const float* saved, float slope, const float* W0, const float* b0, const float* W3,
                             const float* b3, const int64_t* labels, const float* weight, float drop_p,
                             const uint64_t* rng_state, float* H3, float* Z, float* logits, float* dlogits, float* dZ,
                             float* dY, float* bn_bwd, float* dgamma, float* dbeta, float* stats, float* ws,
                             const float* bn_part, int bn_tiles, float* saved_out, float* running_mean, float* running_var,
                             float momentum, float eps, int defer, void* H3b, void* Zb, void* dZb, void* dlb, int ldb16,
                             const int32_t* n_dev, const int32_t* label_rows, int lddl, void* stream) {
    ERC_REQUIRE(lddl == 0 || lddl >= C, "head_fused: lddl = %d < C = %d", lddl, C);
    ERC_REQUIRE((!H3b && !Zb && !dZb && !dlb) || (H3b && Zb && dZb && dlb && ldb16 >= F && ldb16 % 4 == 0 && (((uintptr_t)H3b) & 7) == 0),
                "head_fused: bf16 operand copies (all four or none, pitch %% 4 == 0)");
    // (H3, Z, dZ, dlogits may be NULL when their bf16 copies are taken: the weight-gradient launch reads those, nothing reads the fp32 ones)
    ERC_REQUIRE(H2 && gamma && beta && (saved || bn_part) && W0 && b0 && W3 && b3 && labels && logits && dY && bn_bwd && dgamma && dbeta &&
                    stats && ws && ((H3 && Z && dZ && dlogits) || (H3b && !H3 && !Z && !dZ && !dlogits)),
                "head_fused: null pointer (H3 / Z / dZ / dlogits: all four, or none of them next to the bf16 copies)");
    ERC_REQUIRE(n_rows > 0 && F >= 4 && F <= HF_MAXF && F % 4 == 0 && C > 0 && C <= HF_MAXC && ldh >= F && ldh % 4 == 0,
                "head_fused: n_rows=%d F=%d C=%d ldh=%d unsupported (F <= %d, F %% 4 == 0, C <= %d)", n_rows, F, C, ldh,
                HF_MAXF, HF_MAXC);
    ERC_REQUIRE(al16(H2) && al16(gamma) && al16(beta) && (!saved || al16(saved)) && al16(W0) && (!H3 || al16(H3)), "head_fused: 16-byte alignment");
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state), "head_fused: drop_p=%f", (double)drop_p);
    ERC_REQUIRE(!bn_part || (bn_tiles > 0 && saved_out && running_mean && running_var), "head_fused: BatchNorm partials operands");
    HeadP p{};
    p.H2 = H2, p.gamma = gamma, p.beta = beta, p.saved = saved, p.W0 = W0, p.b0 = b0, p.W3 = W3, p.b3 = b3;
    p.labels = labels, p.weight = weight, p.rng = rng_state, p.H3 = H3, p.Z = Z, p.logits = logits, p.dlogits = dlogits;
    p.dZ = dZ, p.dY = dY, p.bn_bwd = bn_bwd, p.dgamma = dgamma, p.dbeta = dbeta, p.stats = stats;
    p.slope = slope, p.drop_p = drop_p, p.ldh = ldh, p.N = n_rows, p.F = F, p.C = C;
    p.bn_part = bn_part, p.bn_tiles = bn_tiles, p.saved_out = saved_out, p.running_mean = running_mean, p.running_var = running_var;
    p.momentum = momentum, p.eps = eps;
    p.defer = defer ? 1 : 0;
    p.H3b = (unsigned short*)H3b, p.Zb = (unsigned short*)Zb, p.dZb = (unsigned short*)dZb, p.dlb = (unsigned short*)dlb, p.ldb16 = ldb16;
    p.n_dev = n_dev, p.label_rows = label_rows, p.lddl = lddl ? lddl : C;
    p.stamps = g_head_stamps;
    static int wave_form = -1;
    if (wave_form < 0) {
        const char* e = getenv("ERC_HEAD_WAVE");
        wave_form = e ? atoi(e) : 1;
    }
    if (!bn_part && !defer && n_rows > 8192 && wave_form) {
        // throughput form: one wavefront per row tile, workgroups of four looping over the tiles, two per CU
        const int tiles = erc_cdiv(n_rows, 16);
        const int grid_w = erc_cdiv(tiles, 4) < 512 ? erc_cdiv(tiles, 4) : 512;
        p.part = ws;
        p.counter = reinterpret_cast<int*>(ws + (int64_t)grid_w * HF_PART);
        hipLaunchKernelGGL(head_rows_kernel, dim3(grid_w), dim3(256), 0, (hipStream_t)stream, p);
        ERC_LAUNCH_CHECK("head_fused (rows)");
        return ERC_OK;
    }
    const int rpw = erc_head_fused_rows_per_workgroup(n_rows) / 16;
    const int grid = erc_cdiv(n_rows, 16 * rpw);
    p.part = ws;
    p.counter = reinterpret_cast<int*>(ws + (int64_t)grid * HF_PART);
    if (rpw == 2) {
        if (bn_part) hipLaunchKernelGGL((head_fused_kernel<true, 2>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((head_fused_kernel<false, 2>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    } else {
        if (bn_part) hipLaunchKernelGGL((head_fused_kernel<true, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((head_fused_kernel<false, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    }
    ERC_LAUNCH_CHECK("head_fused");
    return ERC_OK;
}

extern "C" int erc_head_fused(const float* H2, int ldh, int n_rows, int F, int C, const float* gamma, const float* beta,
                              const float* saved, float slope, const float* W0, const float* b0, const float* W3,
                              const float* b3, const int64_t* labels, const float* weight, float drop_p,
                              const uint64_t* rng_state, float* H3, float* Z, float* logits, float* dlogits, float* dZ,
                              float* dY, float* bn_bwd, float* dgamma, float* dbeta, float* stats, float* ws, void* H3b, void* Zb,
                              void* dZb, void* dlb, int ldb16, const int32_t* n_dev, const int32_t* label_rows, int lddl, void* stream) {
    ERC_REQUIRE(saved, "head_fused: null pointer");
    return head_fused_launch(H2, ldh, n_rows, F, C, gamma, beta, saved, slope, W0, b0, W3, b3, labels, weight, drop_p, rng_state,
                             H3, Z, logits, dlogits, dZ, dY, bn_bwd, dgamma, dbeta, stats, ws, nullptr, 0, nullptr, nullptr,
                             nullptr, 0.f, 0.f, 0, H3b, Zb, dZb, dlb, ldb16, n_dev, label_rows, lddl, stream);
}

extern "C" int erc_head_fused_part_floats(void) { return HF_PART; }
// diagnostic: resident workgroups per CU of the throughput form (head_rows_kernel), by the runtime's occupancy query
extern "C" int erc_head_rows_occupancy(void) {
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, head_rows_kernel, 256, 0) != hipSuccess) return -1;
    return n;
}

// erc_head_fused with BatchNorm's batch statistics finalised inside: bn_part [bn_tiles][2F] = the per-tile column sums
// (sum x | sum x^2) erc_cogmen_fwd_tile leaves in bn mode 2; saved [2F] (mean | rstd) becomes an OUTPUT, the running
// statistics are updated (momentum, unbiased variance) as nn.BatchNorm1d does in training mode.
extern "C" int erc_head_fused_bn(const float* H2, int ldh, int n_rows, int F, int C, const float* gamma, const float* beta,
                                 float* saved, float slope, const float* W0, const float* b0, const float* W3,
                                 const float* b3, const int64_t* labels, const float* weight, float drop_p,
                                 const uint64_t* rng_state, float* H3, float* Z, float* logits, float* dlogits, float* dZ,
                                 float* dY, float* bn_bwd, float* dgamma, float* dbeta, float* stats, float* ws,
                                 const float* bn_part, int bn_tiles, float* running_mean, float* running_var, float momentum,
                                 float eps, int defer_reduce, void* H3b, void* Zb, void* dZb, void* dlb, int ldb16, const int32_t* n_dev,
                                 const int32_t* label_rows, int lddl, void* stream) {
    ERC_REQUIRE(bn_part && saved, "head_fused_bn: null pointer");
    return head_fused_launch(H2, ldh, n_rows, F, C, gamma, beta, nullptr, slope, W0, b0, W3, b3, labels, weight, drop_p, rng_state,
                             H3, Z, logits, dlogits, dZ, dY, bn_bwd, dgamma, dbeta, stats, ws, bn_part, bn_tiles, saved,
                             running_mean, running_var, momentum, eps, defer_reduce, H3b, Zb, dZb, dlb, ldb16, n_dev, label_rows, lddl, stream);
}

extern "C" int erc_bn_bwd_apply(const float* x, int ldx, int N, int F, const float* gamma, const float* saved,
                                const float* bn_bwd, const float* dY, int lddy, float* dx, int lddx, void* stream) {
    ERC_REQUIRE(x && gamma && saved && bn_bwd && dY && dx, "bn_bwd_apply: null pointer");
    ERC_REQUIRE(N > 0 && F >= 4 && F % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0,
                "bn_bwd_apply: N=%d F=%d pitches %d %d %d", N, F, ldx, lddy, lddx);
    ERC_REQUIRE(al16(x) && al16(gamma) && al16(saved) && al16(bn_bwd) && al16(dY) && al16(dx), "bn_bwd_apply: 16-byte alignment");
    const int64_t total = (int64_t)N * (F / 4);
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, N, F, gamma, saved, bn_bwd,
                       dY, lddy, dx, lddx);
    ERC_LAUNCH_CHECK("bn_bwd_apply");
    return ERC_OK;
}
