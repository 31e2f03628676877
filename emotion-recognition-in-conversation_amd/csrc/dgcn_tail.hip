// DialogueGCN, everything between the RGCN layer's partial outputs and the gradient that goes back into it, as ONE launch:
//   Hc = sum of the RGCN slabs + bias                                    (models/rgcn.py:345-355)
//   GraphConv: AGG_i = sum_{j -> i} Hc_j ; graph_out = W_rel AGG + b + W_root Hc      (dgcn_models.py:42,46)
//   Classifier on [features | graph_out]: Zc = dropout(relu(lin1 x)), logits = lin2 Zc  (dgcn_models.py:163-170)
//   (class-weighted) cross entropy, its gradient, dZc, dXc = dZc W1                     (dgcn.py:124)
//   GraphConv backward, the row-local part: dAGG = dG W_rel, dHc = dG W_root            (dG = dXc[:, 200:])
// Before: slab_reduce + csr_sum + 2 GEMMs + GEMM + head_ce + 3 GEMMs = 9 launches of 5 - 14 us for N ~ 600 rows (launch floors:
// 65 us of the 417 us step).  Here a workgroup owns 16 rows; the only rows it needs from others are the Hc rows of its window
// (+- 10 utterances), which it sums from the slabs itself.  What is left outside is the scatter dHc_j += sum_{j -> i} dAGG_i
// (erc_csr_sum), which needs every workgroup's dAGG.
//
// All products are v_mfma_f32_16x16x4_f32 (exact fp32 multiply-adds, as the GEMMs they replace).  Operand access: the matrix
// core sums over k whichever k sits in which slot, so k-step j of a group of 16 takes k = 16 S + 4 (lane >> 4) + j for A and B
// alike -- a lane's four steps are ONE 16-byte load (LDS for the row tile, global / L2 for a weight row).
#include "erc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TG = 200, TH = 100, TX = 300;     // features, hidden width, classifier input
constexpr int TR = 16;                          // rows of a workgroup
constexpr int THL = 10;                         // window the kernel is built for (wp, wf <= 10)
constexpr int TW = TR + 2 * THL;                // 36 window rows
constexpr int PX = 308;                         // LDS pitch of the [features | graph_out] tile (K = 304 used)
constexpr int PH = 116;                         // LDS pitch of the 100-wide tiles (K = 112 used)
constexpr int PD = 20;                          // LDS pitch of the dlogits tile (K = 16 used)
constexpr int TMAXC = 8;
constexpr int TNT = 7;                          // 16-column tiles over 100

struct TailP {
    const float* slabs; int n_slabs; int64_t slab_stride;     // RGCN partial outputs [S][N * 100]
    const float* rgcn_bias;
    const int32_t* in_ptr; const int32_t* in_src;             // CSR by target
    const float* W_rel; const float* b_rel; const float* W_root;   // [100,100] each, [out][in]
    const float* W1; const float* b1;                         // [100,300], [100]
    const float* W2; const float* b2;                         // [C,100], [C]
    const int64_t* labels; const float* weight;               // [N]; class weights [C] or null
    const uint64_t* rng;                                      // {offset, seed}: read when drop_p > 0
    float drop_p;
    int N, C;
    float* Xc; int ldx;                                       // [N,300]: columns [0,200) in, [200,300) out
    float* Hc; float* AGG; float* Zc;                         // [N,100] out
    float* logits; float* dlogits;                            // [N,C] out
    float* dZc; float* dXc; int lddx; float* dAGG; float* dHc;
    float* stats;                                             // [0] loss [1] hits [2] weight sum; [4] arrival; [8 + 2 g] partials
    uint64_t* stamps;                                         // diagnostic (erc_dgcn_tail_set_stamps): phase stamps of workgroup 0
};
uint64_t* g_tail_stamps = nullptr;
#define TAIL_STAMP(slot)                                                                             \
    do {                                                                                             \
        if (p.stamps && blockIdx.x == 0 && tid == 0) p.stamps[slot] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)


__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// B operand of one group for a weight stored [n][k] (nn.Linear layout, the product x W^T): 16 bytes of row n
__device__ __forceinline__ float4 wrow4(const float* __restrict__ W, int ldw, int n, bool nv, int k0, int K) {
    // (rows are 16-byte aligned: ldw % 4 == 0; K % 4 == 0, so a quad is inside the row or outside)
    const bool v = nv && k0 < K;
    const float4 w = *reinterpret_cast<const float4*>(W + (int64_t)n * ldw + (v ? k0 : 0));
    const float m = v ? 1.f : 0.f;
    return make_float4(w.x * m, w.y * m, w.z * m, w.w * m);
}
// ... for a weight stored [k][n] (the product dy W): four rows, one element each
__device__ __forceinline__ float4 wcol4(const float* __restrict__ W, int ldw, int n, bool nv, int k0, int K) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = nv && k0 + j < K;
        v[j] = W[(int64_t)(ok ? k0 + j : 0) * ldw + n] * (ok ? 1.f : 0.f);
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// one 16 x 16 output tile over NG groups of 16 k: A rows from LDS (pitch pa, K padded with zeros), B fragments from registers
template <int NG>
__device__ __forceinline__ f32x4 tile_product(const float* sA, int pa, int l15, int kq, const float4 (&b)[NG], f32x4 acc) {
#pragma unroll
    for (int S = 0; S < NG; ++S) {
        const float4 a = *reinterpret_cast<const float4*>(sA + l15 * pa + 16 * S + 4 * kq);
        acc = mfma4(a, b[S], acc);
    }
    return acc;
}

constexpr int TNW = 8;              // wavefronts of a workgroup
constexpr int TNTH = 64 * TNW;

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load in flight, and the weight
// fragments of the NEXT phase are requested before each barrier (a phase would otherwise start with an L2 round trip per tile:
// measured 50 us for the launch with four wavefronts and loads issued per tile, against ~6 us of matrix-core time)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(TNTH) void dgcn_tail_kernel(const TailP p) {
    __shared__ __attribute__((aligned(16))) float sX[TR * PX];      // [features | graph_out | 0]
    __shared__ __attribute__((aligned(16))) float sHw[TW * PH];     // Hc of the window rows
    __shared__ __attribute__((aligned(16))) float sAGG[TR * PH];
    __shared__ __attribute__((aligned(16))) float sHo[TR * PH];     // Hc of the own rows (zero padded)
    __shared__ __attribute__((aligned(16))) float sZ[TR * PH];
    __shared__ __attribute__((aligned(16))) float sD[TR * PD];
    // (static LDS is limited to 64 KB: tiles whose lifetimes do not overlap share their space; the K padding stays zero, nobody
    //  writes columns >= 100)
    float* const sdZ = sAGG;                                        // dZc: after the GraphConv product has read AGG
    float* const sdG = sHo;                                         // dG:  likewise
    float (*const sPart)[TR][17] = reinterpret_cast<float (*)[TR][17]>(sHw);     // lin2 partials: after the AGG sums have read Hc
    __shared__ double s_red[TNTH];
    __shared__ double s_wsum;
    __shared__ float s_invw;
    __shared__ float s_ce[4][2];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, C = p.C;
    const int r0 = blockIdx.x * TR;
    TAIL_STAMP(0);
    const int w0 = max(0, r0 - THL), w1 = min(N, r0 + TR + THL);
    // the 16-column tile of the 100-wide products this wavefront owns (wavefront 7: none)
    const int n1 = 16 * wave + l15;
    const bool t1 = wave < TNT, n1v = t1 && n1 < TH;
    const int n1c = min(n1, TH - 1);

    // ---- requests, in the order they are needed (a wavefront's loads complete in order): the tile's CSR bounds, the feature
    //      columns and lin1's weight fragments for them (that part of lin1 does not wait for the graph), then the slabs
    int my_ptr = 0;
    if (tid <= TR) my_ptr = p.in_ptr[min(r0 + tid, N)];
    float4 xv[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {      // 16 x 77 quads of the [features | graph_out | 0] tile
        const int x = tid + TNTH * u, i = min(x / (PX / 4), TR - 1), c4 = x % (PX / 4);
        xv[u] = *reinterpret_cast<const float4*>(p.Xc + (int64_t)min(r0 + i, N - 1) * p.ldx + 4 * min(c4, TG / 4 - 1));
    }
    constexpr int NGF = 12;            // groups of 16 k that lie inside the 200 feature columns
    float4 bL1[19], bRel[7], bRoot[7];
#pragma unroll
    for (int S = 0; S < NGF; ++S) bL1[S] = wrow4(p.W1, TX, n1c, n1v, 16 * S + 4 * kq, TX);
    // window row i, column quad c4 of items x = tid and tid + 512 (900 items); up to 8 slabs per pass, all requested at once
    float4 sv[2][8];
    const int n_sl = p.n_slabs;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int x = tid + TNTH * u, i = min(x / (TH / 4), TW - 1), c4 = x % (TH / 4);
        const float* src = p.slabs + (int64_t)min(w0 + i, N - 1) * TH + 4 * c4;
#pragma unroll
        for (int s = 0; s < 8; ++s) sv[u][s] = *reinterpret_cast<const float4*>(src + min(s, n_sl - 1) * p.slab_stride);
    }
    int* const sPtr = reinterpret_cast<int*>(sD);                    // [17]  (the dlogits tile is written much later)
    int* const sSrc = reinterpret_cast<int*>(sZ);                    // in-edge sources of the tile's rows (<= 16 x 21; Zc: later)
    if (tid <= TR) sPtr[tid] = my_ptr;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int x = tid + TNTH * u, i = x / (PX / 4), c4 = x % (PX / 4);
        if (x < TR * (PX / 4)) {
            const bool v = r0 + i < N && 4 * c4 < TG;
            *reinterpret_cast<float4*>(sX + i * PX + 4 * c4) = v ? xv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // the GraphConv weight fragments, the biases, the labels of the rows this wavefront scores
#pragma unroll
    for (int S = 0; S < 7; ++S) {
        bRel[S] = wrow4(p.W_rel, TH, n1c, n1v, 16 * S + 4 * kq, TH);
        bRoot[S] = wrow4(p.W_root, TH, n1c, n1v, 16 * S + 4 * kq, TH);
    }
    const float bias_rel = p.b_rel[n1c], bias_1 = p.b1[n1c], bias_2 = p.b2[min(l15, C - 1)];
    // cross entropy: wavefront w < 4 scores rows 4 (lane >> 4) + w
    const int ce_row = 4 * kq + min(wave, 3);
    const bool ce_rv = r0 + ce_row < N;
    const int ce_y = (int)p.labels[min(r0 + ce_row, N - 1)];
    lds_barrier();
    TAIL_STAMP(1);
    const int e_lo = sPtr[0], n_e = min(sPtr[min(TR, N - r0)] - e_lo, TR * PH);
    int my_src = 0;
    if (tid < n_e) my_src = p.in_src[e_lo + tid];
    // ---- lin1, the feature columns (while the edge list travels)
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    if (t1) {
#pragma unroll
        for (int S = 0; S < NGF; ++S) {
            const float4 a = *reinterpret_cast<const float4*>(sX + l15 * PX + 16 * S + 4 * kq);
            acc1 = mfma4(a, bL1[S], acc1);
        }
    }
#pragma unroll
    for (int S = NGF; S < 19; ++S) bL1[S] = wrow4(p.W1, TX, n1c, n1v, 16 * S + 4 * kq, TX);
    // ---- Hc = sum of the slabs + bias (slab order, bias last: as erc_slab_reduce)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int x = tid + TNTH * u, i = x / (TH / 4), c4 = x % (TH / 4);
        if (x < TW * (TH / 4)) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w0 + i < w1) {
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < n_sl) acc.x += sv[u][s].x, acc.y += sv[u][s].y, acc.z += sv[u][s].z, acc.w += sv[u][s].w;
                for (int s = 8; s < n_sl; ++s) {      // (more than 8 partial outputs: relation space with a long K split)
                    const float4 v = *reinterpret_cast<const float4*>(p.slabs + s * p.slab_stride + (int64_t)(w0 + i) * TH + 4 * c4);
                    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
                }
                const float4 b = *reinterpret_cast<const float4*>(p.rgcn_bias + 4 * c4);
                acc.x += b.x, acc.y += b.y, acc.z += b.z, acc.w += b.w;
                if (w0 + i >= r0 && w0 + i < r0 + TR) *reinterpret_cast<float4*>(p.Hc + (int64_t)(w0 + i) * TH + 4 * c4) = acc;
            }
            *reinterpret_cast<float4*>(sHw + i * PH + 4 * c4) = acc;
        }
    }
    for (int x = tid; x < TR * (PH - TH); x += TNTH) {       // K padding of the 100-wide A tiles
        const int i = x / (PH - TH), c = TH + x % (PH - TH);
        sAGG[i * PH + c] = 0.f, sHo[i * PH + c] = 0.f;
    }
    if (tid < n_e) sSrc[tid] = my_src;
    for (int x = tid + TNTH; x < n_e; x += TNTH) sSrc[x] = p.in_src[e_lo + x];
    // the weight sum of the class-weighted mean (every workgroup: N labels, no exchange)
    if (p.weight) {
        double wacc = 0.0;
        for (int i = tid; i < N; i += TNTH) wacc += (double)p.weight[p.labels[i]];
        s_red[tid] = wacc;
    }
    lds_barrier();
    TAIL_STAMP(2);
    if (p.weight) {
        for (int o = TNTH / 2; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            lds_barrier();
    TAIL_STAMP(3);
        }
        if (tid == 0) s_wsum = s_red[0], s_invw = (float)(1.0 / s_red[0]);
    } else if (tid == 0) {
        s_wsum = (double)N, s_invw = (float)(1.0 / (double)N);
    }
    // ---- AGG_i = sum over the in-edges' source rows (LDS only); the own Hc rows as an A tile
    for (int x = tid; x < TR * (TH / 4); x += TNTH) {
        const int i = x / (TH / 4), c4 = x % (TH / 4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), own = acc;
        if (r0 + i < N) {
            const int e0 = sPtr[i] - e_lo, e1 = min(sPtr[i + 1] - e_lo, n_e);
            for (int e = e0; e < e1; ++e) {
                const int j = min(max(sSrc[e] - w0, 0), TW - 1);
                const float4 v = *reinterpret_cast<const float4*>(sHw + j * PH + 4 * c4);
                acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
            }
            own = *reinterpret_cast<const float4*>(sHw + (r0 + i - w0) * PH + 4 * c4);
            *reinterpret_cast<float4*>(p.AGG + (int64_t)(r0 + i) * TH + 4 * c4) = acc;
        }
        *reinterpret_cast<float4*>(sAGG + i * PH + 4 * c4) = acc;
        *reinterpret_cast<float4*>(sHo + i * PH + 4 * c4) = own;
    }
    lds_barrier();
    TAIL_STAMP(4);
    for (int x = tid; x < TR * (PH - TH); x += TNTH) sZ[(x / (PH - TH)) * PH + TH + x % (PH - TH)] = 0.f;   // (the edge list is done)
    if (tid < TR * PD) sD[tid] = 0.f;
    const float inv_w = s_invw;

    // ---- GraphConv: graph_out = AGG W_rel^T + b + Hc W_root^T
    if (t1) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_product<7>(sAGG, PH, l15, kq, bRel, acc);
        acc = tile_product<7>(sHo, PH, l15, kq, bRoot, acc);
        const float bn = bias_rel;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            const float v = acc[r] + bn;
            if (n1v) sX[row * PX + TG + n1] = v;
            if (n1v && r0 + row < N) p.Xc[(int64_t)(r0 + row) * p.ldx + TG + n1] = v;
        }
    }
    // requests for the phases behind lin1: lin2's K group of this wavefront, dlogits W2, the W1 columns of dXc
    const bool cv2 = l15 < C;
    const float4 bL2 = wrow4(p.W2, TH, min(l15, C - 1), t1 && cv2, 16 * wave + 4 * kq, TH);
    const float4 bDz = wcol4(p.W2, TH, n1c, n1v, 4 * kq, C);
    float4 bDx[3][7];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int n = 16 * (wave + TNW * u) + l15;
        const bool nv = wave + TNW * u < 19 && n < TX;
#pragma unroll
        for (int S = 0; S < 7; ++S) bDx[u][S] = wcol4(p.W1, TX, min(n, TX - 1), nv, 16 * S + 4 * kq, TH);
    }
    lds_barrier();
    TAIL_STAMP(5);

    // ---- lin1 + ReLU + dropout
    const float p_drop = p.drop_p;
    const float keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    uint64_t roff = 0, rseed = 0;
    if (p_drop > 0.f) roff = p.rng[0], rseed = p.rng[1];
    if (t1) {
        f32x4 acc = acc1;
#pragma unroll
        for (int S = NGF; S < 19; ++S) {       // the graph_out columns (and the last 8 feature columns)
            const float4 a = *reinterpret_cast<const float4*>(sX + l15 * PX + 16 * S + 4 * kq);
            acc = mfma4(a, bL1[S], acc);
        }
        const float bn = bias_1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            float v = fmaxf(acc[r] + bn, 0.f);
            if (p_drop > 0.f) {       // uniform
                const float u = erc_uniform(rseed, roff, (uint64_t)(r0 + row) * TH + n1c);
                v = u >= p_drop ? v * keep : 0.f;
            }
            if (n1v) sZ[row * PH + n1] = v;
            if (n1v && r0 + row < N) p.Zc[(int64_t)(r0 + row) * TH + n1] = v;
        }
    }
    // requests: the W_rel / W_root columns of the GraphConv backward (14 tiles: this wavefront's t and t + 8)
    float4 bGb[2][7];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = wave + TNW * u;
        const bool root = t >= TNT;
        const int n = 16 * (root ? t - TNT : t) + l15;
        const bool nv = t < 2 * TNT && n < TH;
        const float* W = root ? p.W_root : p.W_rel;
#pragma unroll
        for (int S = 0; S < 7; ++S) bGb[u][S] = wcol4(W, TH, min(n, TH - 1), nv, 16 * S + 4 * kq, TH);
    }
    lds_barrier();
    TAIL_STAMP(6);

    // ---- lin2 (K split over the wavefronts), cross entropy and its gradient
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t1) {
            const float4 a = *reinterpret_cast<const float4*>(sZ + l15 * PH + 16 * wave + 4 * kq);
            acc = mfma4(a, bL2, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sPart[wave][4 * kq + r][l15] = acc[r];
    }
    lds_barrier();
    TAIL_STAMP(7);
    int ticket = -1;
    if (wave < 4) {
        const bool cv = l15 < C;
        const int row = ce_row, y = ce_y;
        const bool rv = ce_rv;
        float lg = bias_2;
#pragma unroll
        for (int w = 0; w < TNT; ++w) lg += sPart[w][row][l15];
        float mx = cv ? lg : -INFINITY;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        int am = (cv && lg == mx) ? l15 : 99;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) am = min(am, __shfl_xor(am, o, 64));
        float se = cv ? expf(lg - mx) : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
        const float lse = mx + logf(se);
        float ly = (cv && l15 == y) ? lg : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ly += __shfl_xor(ly, o, 64);
        const float wy = p.weight ? p.weight[y] : 1.f;
        const float d = (cv && rv) ? wy * inv_w * (expf(lg - lse) - (l15 == y ? 1.f : 0.f)) : 0.f;
        sD[row * PD + l15] = d;
        if (cv && rv) {
            p.logits[(int64_t)(r0 + row) * C + l15] = lg;
            p.dlogits[(int64_t)(r0 + row) * C + l15] = d;
        }
        // lanes 0, 16, 32, 48 hold the four row groups' parts
        float lacc = (rv && l15 == 0) ? wy * (lse - ly) : 0.f, hits = (rv && l15 == 0 && am == y) ? 1.f : 0.f;
        lacc += __shfl_xor(lacc, 16, 64), hits += __shfl_xor(hits, 16, 64);
        lacc += __shfl_xor(lacc, 32, 64), hits += __shfl_xor(hits, 32, 64);
        if (lane == 0) s_ce[wave][0] = lacc, s_ce[wave][1] = hits;
    }
    lds_barrier();
    TAIL_STAMP(8);
    // the workgroup's part of the loss / accuracy and its arrival ticket: requested here by the wavefront with the least to do,
    // looked at when the kernel ends (the fence in front of the ticket waits for the wavefront's own stores only)
    if (tid == 64 * (TNW - 1)) {
        const float lacc = (s_ce[0][0] + s_ce[1][0]) + (s_ce[2][0] + s_ce[3][0]), hits = (s_ce[0][1] + s_ce[1][1]) + (s_ce[2][1] + s_ce[3][1]);
        __hip_atomic_store(p.stats + 8 + 2 * blockIdx.x, lacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.stats + 9 + 2 * blockIdx.x, hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        ticket = atomicAdd(reinterpret_cast<int*>(p.stats + 4), 1);
    }

    // ---- dZc = (dlogits W2) through the ReLU / dropout mask
    if (t1) {
        const float4 a = *reinterpret_cast<const float4*>(sD + l15 * PD + 4 * kq);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma4(a, bDz, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            const float z = n1v ? sZ[row * PH + n1] : 0.f;
            const float v = z > 0.f ? acc[r] * keep : 0.f;
            if (n1v) sdZ[row * PH + n1] = v;
            if (n1v && r0 + row < N) p.dZc[(int64_t)(r0 + row) * TH + n1] = v;
        }
    }
    lds_barrier();
    TAIL_STAMP(9);

    // ---- dXc = dZc W1 (300 columns; the last 100 are dG, the gradient of the GraphConv output)
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = wave + TNW * u;
        if (t < 19) {
            const int n = 16 * t + l15;
            const bool nv = n < TX;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_product<7>(sdZ, PH, l15, kq, bDx[u], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r;
                if (nv && n >= TG) sdG[row * PH + n - TG] = acc[r];
                if (nv && r0 + row < N) p.dXc[(int64_t)(r0 + row) * p.lddx + n] = acc[r];
            }
        }
    }
    lds_barrier();
    TAIL_STAMP(10);

    // ---- GraphConv backward, row-local: dAGG = dG W_rel ; dHc = dG W_root (the scatter over the out-edges follows outside)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = wave + TNW * u;
        if (t < 2 * TNT) {
            const bool root = t >= TNT;
            const int n = 16 * (root ? t - TNT : t) + l15;
            const bool nv = n < TH;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_product<7>(sdG, PH, l15, kq, bGb[u], acc);
            float* out = root ? p.dHc : p.dAGG;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r;
                if (nv && r0 + row < N) out[(int64_t)(r0 + row) * TH + n] = acc[r];
            }
        }
    }

    TAIL_STAMP(15);
    // ---- loss / accuracy: the last arriver combines the workgroups' parts in workgroup order
    if (tid == 64 * (TNW - 1)) s_last = (ticket == (int)gridDim.x - 1);
    __syncthreads();
    if (s_last) {
        __threadfence();
        double l = 0.0, h = 0.0;
        for (int g = tid; g < (int)gridDim.x; g += TNTH) {
            l += (double)__hip_atomic_load(p.stats + 8 + 2 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h += (double)__hip_atomic_load(p.stats + 9 + 2 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        s_red[tid] = l;
        __syncthreads();
        for (int o = TNTH / 2; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        const double lsum = s_red[0];
        __syncthreads();
        s_red[tid] = h;
        __syncthreads();
        for (int o = TNTH / 2; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        if (tid == 0) {
            p.stats[0] = (float)(lsum / s_wsum);
            p.stats[1] = (float)s_red[0];
            p.stats[2] = (float)s_wsum;
            *reinterpret_cast<int*>(p.stats + 4) = 0;
        }
    }
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

}  // namespace

// diagnostic: 16 x uint64 phase stamps (10 ns ticks) of workgroup 0 of the following launches; NULL = off
extern "C" int erc_dgcn_tail_set_stamps(uint64_t* stamps) {
    g_tail_stamps = stamps;
    return ERC_OK;
}
extern "C" int erc_dgcn_tail_max_rows(void) { return 8192; }
extern "C" int erc_dgcn_tail_max_window(void) { return THL; }
extern "C" int64_t erc_dgcn_tail_stats_floats(int n_rows) { return 16 + 2 * (int64_t)erc_cdiv(n_rows, TR); }

extern "C" int erc_dgcn_tail(const float* slabs, int n_slabs, int64_t slab_stride, const float* rgcn_bias, const int32_t* in_ptr,
                             const int32_t* in_src, int window, const float* W_rel, const float* b_rel, const float* W_root,
                             const float* W1, const float* b1, const float* W2, const float* b2, const int64_t* labels,
                             const float* weight, int n_classes, int n_rows, float drop_p, const uint64_t* rng, float* Xc, int ldx,
                             float* Hc, float* AGG, float* Zc, float* logits, float* dlogits, float* dZc, float* dXc, int lddx,
                             float* dAGG, float* dHc, float* stats, void* stream) {
    ERC_REQUIRE(slabs && rgcn_bias && in_ptr && in_src && W_rel && b_rel && W_root && W1 && b1 && W2 && b2 && labels && Xc && Hc &&
                    AGG && Zc && logits && dlogits && dZc && dXc && dAGG && dHc && stats, "dgcn_tail: null pointer");
    ERC_REQUIRE(n_rows > 0 && n_rows <= erc_dgcn_tail_max_rows() && n_classes > 0 && n_classes <= TMAXC && n_slabs >= 1,
                "dgcn_tail: n_rows=%d n_classes=%d n_slabs=%d unsupported", n_rows, n_classes, n_slabs);
    ERC_REQUIRE(window >= 0 && window <= THL, "dgcn_tail: window %d (the kernel holds %d rows either side)", window, THL);
    ERC_REQUIRE(ldx >= TX && lddx >= TX && ldx % 4 == 0 && slab_stride % 4 == 0, "dgcn_tail: ldx=%d lddx=%d slab_stride=%lld", ldx, lddx,
                (long long)slab_stride);
    ERC_REQUIRE(al16(slabs) && al16(rgcn_bias) && al16(W_rel) && al16(W_root) && al16(W1) && al16(W2) && al16(Xc) && al16(Hc) &&
                    al16(AGG), "dgcn_tail: operands must be 16-byte aligned");
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "dgcn_tail: drop_p=%f", (double)drop_p);
    TailP p;
    p.slabs = slabs; p.n_slabs = n_slabs; p.slab_stride = slab_stride; p.rgcn_bias = rgcn_bias; p.in_ptr = in_ptr; p.in_src = in_src;
    p.W_rel = W_rel; p.b_rel = b_rel; p.W_root = W_root; p.W1 = W1; p.b1 = b1; p.W2 = W2; p.b2 = b2; p.labels = labels; p.weight = weight;
    p.rng = rng; p.drop_p = drop_p; p.N = n_rows; p.C = n_classes; p.Xc = Xc; p.ldx = ldx; p.Hc = Hc; p.AGG = AGG; p.Zc = Zc;
    p.logits = logits; p.dlogits = dlogits; p.dZc = dZc; p.dXc = dXc; p.lddx = lddx; p.dAGG = dAGG; p.dHc = dHc; p.stats = stats; p.stamps = g_tail_stamps;
    hipLaunchKernelGGL(dgcn_tail_kernel, dim3(erc_cdiv(n_rows, TR)), dim3(TNTH), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("dgcn_tail");
    return ERC_OK;
}
