// MMGCN graph operators (track_mm/mmgcn_models.py:8-39,344-394,493-646).
//
// The reference builds a dense (modalities*N)^2 adjacency on the host (python loops, a CPU-allocated scratch
// block per dialogue) and multiplies it densely in each of the 64 GCNII layers although only the per-dialogue,
// per-modality L x L blocks and the same-utterance cross-modal entries are non-zero.  Here the adjacency is kept
// as exactly that structure:
//     blocks  ADJ [B*M][P][P]   (P = padded max length, multiple of 4), cross CR [B][M*M][P]
// Node rows are ordered modality-major: node (m, b, t) = m*N + node_off[b] + t, like torch.cat([a, v, l]).
// The block products are grouped MFMA GEMMs (gemm.hip); this file holds the remaining small operators.
#include "erc_common.h"

namespace {

constexpr int FD = 200;     // node feature width (n_dim = nhidden = 200)
constexpr int MAXP = 128;   // padded dialogue length supported by the per-dialogue kernels
constexpr float PI_F = 3.14159265358979323846f;
constexpr float SHRINK = 0.99999f;

struct L4 {
    float v[4];
};
__device__ __forceinline__ L4 ld4(const float* row, int lane) {
    L4 r;
#pragma unroll
    for (int u = 0; u < 4; ++u) r.v[u] = lane + 64 * u < FD ? row[lane + 64 * u] : 0.f;
    return r;
}
__device__ __forceinline__ float dt4(const L4& a, const L4& b) {
    return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2] + a.v[3] * b.v[3];
}

// node tables of the time-major MMGCN batch: node i = node_off[b] + t  ->  row t*B + b of the [T,B,.] blocks,
// its dialogue and its speaker id (argmax of the one-hot qmask, mmgcn_models.py:540-541)
__global__ __launch_bounds__(256) void mm_meta_kernel(const int64_t* __restrict__ lengths, const float* __restrict__ qmask,
                                                      int64_t q_st, int64_t q_sb, int S, int B, int32_t* __restrict__ node_off,
                                                      int32_t* __restrict__ node_row, int32_t* __restrict__ node_dlg,
                                                      int32_t* __restrict__ node_spk) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ int red[256];
    int acc = 0;
    for (int i = tid; i < b; i += 256) acc += (int)lengths[i];
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const int off = red[0], L = (int)lengths[b];
    if (tid == 0) {
        node_off[b] = off;
        if (b == B - 1) node_off[B] = off + L;
    }
    for (int t = tid; t < L; t += 256) {
        node_row[off + t] = t * B + b;
        node_dlg[off + t] = b;
        const float* row = qmask + (int64_t)t * q_st + (int64_t)b * q_sb;
        int s = 0;
        float best = row[0];
        for (int c = 1; c < S; ++c)
            if (row[c] > best) best = row[c], s = c;
        node_spk[off + t] = s;
    }
}

// dst[(m_off + i), :] = src[row_map[i], :] (+ emb[spk[i], :])  : simple_batch_graphify (mmgcn_utils.py:5-21) plus
// the speaker-embedding add of mmgcn_models.py:540-545
__global__ __launch_bounds__(256) void flatten_kernel(const float* __restrict__ src, int lds, const int32_t* __restrict__ row_map,
                                                      const float* __restrict__ emb, const int32_t* __restrict__ spk,
                                                      int N, float* __restrict__ dst, int ldd) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    const float* s = src + (int64_t)row_map[i] * lds;
    const float* e = emb ? emb + (int64_t)spk[i] * FD : nullptr;
    for (int c = lane; c < FD; c += 64) dst[(int64_t)i * ldd + c] = s[c] + (e ? e[c] : 0.f);
}

// demb[s, :] = sum_{i: spk[i]==s} dl[i, :], fixed order, two stages: EG_CH row chunks per speaker into ws, then their sum.
// Rows are read 8 at a time unconditionally and masked by multiplication.  (One workgroup per speaker walked all N rows
// on 2 of the 256 CUs: 69 us at N = 920; a guarded load per row before that: 152 us.)
constexpr int EG_CH = 32;
__global__ __launch_bounds__(256) void emb_grad_part_kernel(const float* __restrict__ dl, int ld, const int32_t* __restrict__ spk,
                                                             int N, float* __restrict__ part) {
    const int s = blockIdx.x, ch = blockIdx.y, c = threadIdx.x;
    const int cc = min(c, FD - 1);
    const int per = (N + EG_CH - 1) / EG_CH, lo = ch * per, hi = min(N, lo + per);
    float acc = 0.f;
    for (int i0 = lo; i0 < hi; i0 += 8) {
        float v[8], m[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u, N - 1);
            v[u] = dl[(int64_t)i * ld + cc];
            m[u] = (i0 + u < hi && spk[i] == s) ? 1.f : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u] * m[u];
    }
    if (c < FD) part[((int64_t)s * EG_CH + ch) * FD + c] = acc;
}
__global__ __launch_bounds__(256) void emb_grad_sum_kernel(const float* __restrict__ part, float* __restrict__ demb) {
    const int s = blockIdx.x, c = threadIdx.x;
    if (c >= FD) return;
    float v[EG_CH];
#pragma unroll
    for (int ch = 0; ch < EG_CH; ++ch) v[ch] = part[((int64_t)s * EG_CH + ch) * FD + c];
    float acc = 0.f;
#pragma unroll
    for (int ch = 0; ch < EG_CH; ++ch) acc += v[ch];
    demb[s * FD + c] = acc;
}

// xhat = x / |x| ; inv = 1/|x|
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ x, int R, float* __restrict__ xhat,
                                                            float* __restrict__ inv) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= R) return;
    const L4 v = ld4(x + (int64_t)i * FD, lane);
    const float rn = 1.0f / sqrtf(wave_sum(dt4(v, v)));
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < FD) xhat[(int64_t)i * FD + lane + 64 * u] = v.v[u] * rn;
    if (lane == 0) inv[i] = rn;
}

// dx += inv * (dxhat - xhat (xhat . dxhat))
__global__ __launch_bounds__(256) void row_normalize_bwd_kernel(const float* __restrict__ xhat, const float* __restrict__ inv,
                                                                const float* __restrict__ dxhat, int R,
                                                                float* __restrict__ dx) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= R) return;
    const L4 h = ld4(xhat + (int64_t)i * FD, lane), d = ld4(dxhat + (int64_t)i * FD, lane);
    const float dot = wave_sum(dt4(h, d)), rn = inv[i];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < FD) dx[(int64_t)i * FD + lane + 64 * u] += rn * (d.v[u] - h.v[u] * dot);
}

__device__ __forceinline__ float sim_of(float c) { return 1.0f - acosf(SHRINK * c) / PI_F; }
__device__ __forceinline__ float dsim_dc(float c) {
    const float t = SHRINK * c;
    return SHRINK / (PI_F * sqrtf(fmaxf(1.0f - t * t, 1e-12f)));
}

// COS blocks (raw cosines from the grouped GEMM) -> normalised adjacency blocks ADJ, normalised cross entries
// CR[b][m*M+n][p] (m != n), degrees DEG[node], raw cross cosines CCOS.  One wavefront per adjacency row (m, p) of a
// dialogue, two launches: similarities + degrees, then the D^-1/2 . D^-1/2 scaling that needs every degree of the block.
// (One workgroup per dialogue ran the three phases behind barriers on 32 of the 256 CUs: 91 us; and its backward 84 us.)
__global__ __launch_bounds__(256) void adj_rows_kernel(const float* __restrict__ COS, const float* __restrict__ xhat,
                                                       const int32_t* __restrict__ node_off, int M, int N, int P,
                                                       float* __restrict__ ADJ, float* __restrict__ CCOS,
                                                       float* __restrict__ DEG) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, it = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int off = node_off[b], L = node_off[b + 1] - off;
    if (it >= M * L) return;
    const int m = it / L, p = it % L;
    // cross-modal same-utterance similarities of this node
    float cross = 0.f;
    const L4 mine = ld4(xhat + ((int64_t)m * N + off + p) * FD, lane);
    for (int n = 0; n < M; ++n) {
        if (n == m) continue;
        const float c = wave_sum(dt4(mine, ld4(xhat + ((int64_t)n * N + off + p) * FD, lane)));
        if (lane == 0) CCOS[((int64_t)b * M * M + m * M + n) * P + p] = c;
        cross += sim_of(c);
    }
    const float* cr = COS + (((int64_t)b * M + m) * P + p) * P;
    float* ar = ADJ + (((int64_t)b * M + m) * P + p) * P;
    float rs = 0.f;
    for (int q = lane; q < L; q += 64) {
        const float s = sim_of(cr[q]);
        ar[q] = s;
        rs += s;
    }
    rs = wave_sum(rs) + cross;
    if (lane == 0) DEG[(int64_t)m * N + off + p] = rs;
}

__global__ __launch_bounds__(256) void adj_scale_kernel(const float* __restrict__ CCOS, const float* __restrict__ DEG,
                                                        const int32_t* __restrict__ node_off, int M, int N, int P,
                                                        float* __restrict__ ADJ, float* __restrict__ CR) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, it = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int off = node_off[b], L = node_off[b + 1] - off;
    if (it >= M * L) return;
    const int m = it / L, p = it % L;
    const float up = 1.0f / sqrtf(DEG[(int64_t)m * N + off + p]);
    float* ar = ADJ + (((int64_t)b * M + m) * P + p) * P;
    for (int q = lane; q < L; q += 64) ar[q] *= up * (1.0f / sqrtf(DEG[(int64_t)m * N + off + q]));
    if (lane < M && lane != m) {
        const int64_t i_mn = ((int64_t)b * M * M + m * M + lane) * P + p;
        CR[i_mn] = sim_of(CCOS[i_mn]) * up * (1.0f / sqrtf(DEG[(int64_t)lane * N + off + p]));
    }
}

// Backward of adj_finish: from dADJ (blocks) and dCR (directional cross gradients) to
//   G = dCOS + dCOS^T (blocks, so that dXhat_block = G Xhat is one grouped GEMM) and GC (cross, symmetric per
//   unordered pair, applied with cross_apply on Xhat).  Same split: per-row degree gradients DD, then the entries.
__global__ __launch_bounds__(256) void adj_bwd_rows_kernel(const float* __restrict__ COS, const float* __restrict__ CCOS,
                                                           const float* __restrict__ DEG, const float* __restrict__ dADJ,
                                                           const float* __restrict__ dCR,
                                                           const int32_t* __restrict__ node_off, int M, int N, int P,
                                                           float* __restrict__ DD) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, it = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int off = node_off[b], L = node_off[b + 1] - off;
    if (it >= M * L) return;
    const int m = it / L, p = it % L;
    // du_p = sum_q (dA_pq + dA_qp) S_pq u_q  + cross terms;  dd_p = -1/2 du_p d_p^-3/2
    const int64_t base = ((int64_t)b * M + m) * P;
    float acc = 0.f;
    for (int q = lane; q < L; q += 64) {
        const float s = sim_of(COS[(base + p) * P + q]);
        acc += (dADJ[(base + p) * P + q] + dADJ[(base + q) * P + p]) * s * (1.0f / sqrtf(DEG[(int64_t)m * N + off + q]));
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        for (int n = 0; n < M; ++n)
            if (n != m) {
                const float s = sim_of(CCOS[((int64_t)b * M * M + m * M + n) * P + p]);
                // entry ((m,p),(n,p)) and entry ((n,p),(m,p)) both carry u^m_p
                acc += (dCR[((int64_t)b * M * M + m * M + n) * P + p] + dCR[((int64_t)b * M * M + n * M + m) * P + p]) *
                       s * (1.0f / sqrtf(DEG[(int64_t)n * N + off + p]));
            }
        const float u = 1.0f / sqrtf(DEG[(int64_t)m * N + off + p]);
        DD[(int64_t)m * N + off + p] = -0.5f * acc * u * u * u;
    }
}

__global__ __launch_bounds__(256) void adj_bwd_entries_kernel(const float* __restrict__ COS, const float* __restrict__ CCOS,
                                                              const float* __restrict__ DEG, const float* __restrict__ DD,
                                                              const float* __restrict__ dADJ, const float* __restrict__ dCR,
                                                              const int32_t* __restrict__ node_off, int M, int N, int P,
                                                              float* __restrict__ G, float* __restrict__ GC) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, it = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int off = node_off[b], L = node_off[b + 1] - off;
    if (it >= M * L) return;
    const int m = it / L, p = it % L;
    const int64_t base = ((int64_t)b * M + m) * P;
    const float up = 1.0f / sqrtf(DEG[(int64_t)m * N + off + p]), ddp = DD[(int64_t)m * N + off + p];
    // dS_pq = dA_pq u_p u_q + dd_p ; dcos = dS * dsim/dc ; G_pq = dcos_pq + dcos_qp
    for (int q = lane; q < L; q += 64) {
        const float c = COS[(base + p) * P + q];  // symmetric up to rounding; use the (p,q) entry for both halves
        const float uu = up * (1.0f / sqrtf(DEG[(int64_t)m * N + off + q]));
        const float dpq = dADJ[(base + p) * P + q] * uu + ddp;
        const float dqp = dADJ[(base + q) * P + p] * uu + DD[(int64_t)m * N + off + q];
        G[(base + p) * P + q] = (dpq + dqp) * dsim_dc(c);
    }
    if (lane < M && lane != m) {
        const int n = lane;
        const int64_t i_mn = ((int64_t)b * M * M + m * M + n) * P + p, i_nm = ((int64_t)b * M * M + n * M + m) * P + p;
        const float uu = up * (1.0f / sqrtf(DEG[(int64_t)n * N + off + p]));
        // directional entries (m,n) and (n,m) share one cosine: total gradient of that cosine, stored for both
        const float ds = (dCR[i_mn] * uu + ddp) + (dCR[i_nm] * uu + DD[(int64_t)n * N + off + p]);
        GC[i_mn] = ds * dsim_dc(CCOS[i_mn]);
    }
}

// out[(m,p), :] += sum_{n != m} CR[b][m*M+n][p] * h[(n,p), :]   (cross-modal part of A*h, and of its transpose)
__global__ __launch_bounds__(256) void cross_apply_kernel(const float* __restrict__ CR, const float* __restrict__ h, int ldh,
                                                          const int32_t* __restrict__ node_dlg,
                                                          const int32_t* __restrict__ node_off, int M, int N, int P,
                                                          float* __restrict__ out, int ldo) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M * N) return;
    const int m = r / N, i = r % N, b = node_dlg[i], p = i - node_off[b];
    L4 acc = {{0.f, 0.f, 0.f, 0.f}};
    for (int n = 0; n < M; ++n) {
        if (n == m) continue;
        const float c = CR[((int64_t)b * M * M + m * M + n) * P + p];
        const L4 v = ld4(h + ((int64_t)n * N + i) * ldh, lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc.v[u] += c * v.v[u];
    }
    float* o = out + (int64_t)r * ldo;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < FD) o[lane + 64 * u] += acc.v[u];
}

// dCR[b][m*M+n][p] += sum over planes of dhi_pl[(m,p), :] . h_pl[(n,p), :]   (directional); planes = the layers whose
// contributions only meet in this sum (one launch for all of them)
__global__ __launch_bounds__(256) void cross_grad_kernel(const float* __restrict__ dhi, int ldd, const float* __restrict__ h,
                                                         int ldh, const int32_t* __restrict__ node_dlg,
                                                         const int32_t* __restrict__ node_off, int M, int N, int P,
                                                         float* __restrict__ dCR, int planes, int64_t d_plane, int64_t h_plane) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M * N) return;
    const int m = r / N, i = r % N, b = node_dlg[i], p = i - node_off[b];
    for (int n = 0; n < M; ++n) {
        if (n == m) continue;
        float acc = 0.f;
        for (int pl0 = 0; pl0 < planes; pl0 += 4) {      // 4 planes (8 row loads) in flight
            L4 g[4], hh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pl = min(pl0 + u, planes - 1);
                g[u] = ld4(dhi + (int64_t)pl * d_plane + (int64_t)r * ldd, lane);
                hh[u] = ld4(h + (int64_t)pl * h_plane + ((int64_t)n * N + i) * ldh, lane);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += (pl0 + u < planes) ? dt4(g[u], hh[u]) : 0.f;
        }
        const float d = wave_sum(acc);
        if (lane == 0) dCR[((int64_t)b * M * M + m * M + n) * P + p] += d;
    }
}

// GCNII layer tail (mmgcn_models.py:27-39,385-388): out = theta*G + (1-theta)((1-alpha) hi + alpha h0);
// hd = dropout(relu(out))
__global__ __launch_bounds__(256) void gcnii_combine_fwd_kernel(const float* __restrict__ Gm, const float* __restrict__ hi,
                                                                const float* __restrict__ h0, int64_t n, float theta,
                                                                float alpha, float drop_p, const uint64_t* rng,
                                                                uint64_t stream_id, float* __restrict__ hd) {
    uint64_t roff = 0, rseed = 0;
    if (drop_p > 0.f) roff = rng[0], rseed = rng[1] ^ stream_id;
    const float ks = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = Gm[i];
        if (hi) v = theta * v + (1.f - theta) * ((1.f - alpha) * hi[i] + alpha * h0[i]);
        v = fmaxf(v, 0.f);
        if (drop_p > 0.f) v = erc_uniform(rseed, roff, (uint64_t)i) >= drop_p ? v * ks : 0.f;
        hd[i] = v;
    }
}

// backward of the tail: dout = d_hd * [hd > 0] * keep_scale; dG = theta dout; dhi = (1-theta)(1-alpha) dout;
// dh0 += (1-theta) alpha dout.   With hi == nullptr (input layer): dG = dout only.
__global__ __launch_bounds__(256) void gcnii_combine_bwd_kernel(const float* __restrict__ d_hd, const float* __restrict__ hd,
                                                                int64_t n, float theta, float alpha, float keep_scale,
                                                                int plain, float* __restrict__ dG, float* __restrict__ dhi,
                                                                float* __restrict__ dh0, int F, int ld_d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float g = hd[i] > 0.f ? d_hd[i] * keep_scale : 0.f;
        if (plain) {
            dG[i] = g;
        } else {
            const int64_t row = i / F, j = row * ld_d + (i - row * F);   // dhi / dh0 rows have pitch ld_d
            dG[i] = theta * g;
            dhi[j] = (1.f - theta) * (1.f - alpha) * g;
            dh0[j] += (1.f - theta) * alpha * g;
        }
    }
}

// y = dropout(x) elementwise (input dropout of GCNII, mmgcn_models.py:382) and its backward (in place)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t n, float drop_p,
                                                      const uint64_t* rng, uint64_t stream_id, float* __restrict__ y) {
    const uint64_t roff = rng[0], rseed = rng[1] ^ stream_id;
    const float ks = 1.0f / (1.0f - drop_p);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = erc_uniform(rseed, roff, (uint64_t)i) >= drop_p ? x[i] * ks : 0.f;
}

// FE[p, m*400 + c] = relu(dropout(cat[xd, h][(m,p), c]))  (regroup of mmgcn_models.py:570-576 + dropout_/ReLU of
// mmgcn.py:119-120);  backward scatters dFE back to d_xd / d_h.
__global__ __launch_bounds__(256) void regroup_fwd_kernel(const float* __restrict__ xd, const float* __restrict__ hl, int M,
                                                          int N, float drop_p, const uint64_t* rng, uint64_t stream_id,
                                                          float* __restrict__ FE) {
    uint64_t roff = 0, rseed = 0;
    if (drop_p > 0.f) roff = rng[0], rseed = rng[1] ^ stream_id;
    const float ks = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int64_t total = (int64_t)N * M * 2 * FD;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % (2 * FD)), m = (int)((i / (2 * FD)) % M);
        const int64_t p = i / ((int64_t)2 * FD * M);
        const int64_t row = (int64_t)m * N + p;
        float v = c < FD ? xd[row * FD + c] : hl[row * FD + c - FD];
        if (drop_p > 0.f) v = erc_uniform(rseed, roff, (uint64_t)i) >= drop_p ? v * ks : 0.f;
        FE[i] = fmaxf(v, 0.f);
    }
}
__global__ __launch_bounds__(256) void regroup_bwd_kernel(const float* __restrict__ dFE, const float* __restrict__ FE, int M,
                                                          int N, float keep_scale, float* __restrict__ d_xd,
                                                          float* __restrict__ d_h) {
    const int64_t total = (int64_t)N * M * 2 * FD;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % (2 * FD)), m = (int)((i / (2 * FD)) % M);
        const int64_t p = i / ((int64_t)2 * FD * M);
        const int64_t row = (int64_t)m * N + p;
        const float g = FE[i] > 0.f ? dFE[i] * keep_scale : 0.f;
        if (c < FD)
            d_xd[row * FD + c] = g;
        else
            d_h[row * FD + c - FD] = g;
    }
}

// y[i] (+)= a*x[i] masked by (mask[i] != 0) * scale  -- small axpy used to merge gradient streams
__global__ __launch_bounds__(256) void axpy_mask_kernel(const float* __restrict__ x, const float* __restrict__ mask, int64_t n,
                                                        float scale, int accumulate, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = (mask ? (mask[i] != 0.f ? x[i] * scale : 0.f) : x[i] * scale);
        y[i] = accumulate ? y[i] + v : v;
    }
}

int ew_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 2048 ? 2048 : g);
}

}  // namespace

#define NODEG(R) dim3(erc_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream

extern "C" int erc_mm_meta(const int64_t* lengths, const float* qmask, int64_t q_st, int64_t q_sb, int n_speakers, int B,
                           int32_t* node_off, int32_t* node_row, int32_t* node_dlg, int32_t* node_spk, void* stream) {
    ERC_REQUIRE(lengths && qmask && node_off && node_row && node_dlg && node_spk && B > 0 && n_speakers > 0,
                "mm_meta: bad arguments");
    hipLaunchKernelGGL(mm_meta_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, lengths, qmask, q_st, q_sb, n_speakers,
                       B, node_off, node_row, node_dlg, node_spk);
    ERC_LAUNCH_CHECK("mm_meta");
    return ERC_OK;
}
extern "C" int erc_mm_flatten(const float* src, int lds, const int32_t* row_map, const float* emb, const int32_t* spk, int N,
                              float* dst, int ldd, void* stream) {
    ERC_REQUIRE(src && row_map && dst && N > 0 && (!emb || spk), "mm_flatten: bad arguments");
    hipLaunchKernelGGL(flatten_kernel, NODEG(N), src, lds, row_map, emb, spk, N, dst, ldd);
    ERC_LAUNCH_CHECK("mm_flatten");
    return ERC_OK;
}
extern "C" int64_t erc_mm_emb_grad_ws_floats(int n_speakers) { return (int64_t)n_speakers * EG_CH * FD; }
extern "C" int erc_mm_emb_grad(const float* dl, int ld, const int32_t* spk, int N, int n_speakers, float* demb, float* ws,
                               void* stream) {
    ERC_REQUIRE(dl && spk && demb && ws && N > 0 && n_speakers > 0, "mm_emb_grad: bad arguments");
    hipLaunchKernelGGL(emb_grad_part_kernel, dim3(n_speakers, EG_CH), dim3(256), 0, (hipStream_t)stream, dl, ld, spk, N, ws);
    hipLaunchKernelGGL(emb_grad_sum_kernel, dim3(n_speakers), dim3(256), 0, (hipStream_t)stream, ws, demb);
    ERC_LAUNCH_CHECK("mm_emb_grad");
    return ERC_OK;
}
extern "C" int erc_mm_row_normalize(const float* x, int R, float* xhat, float* inv, void* stream) {
    ERC_REQUIRE(x && xhat && inv && R > 0, "mm_row_normalize: bad arguments");
    hipLaunchKernelGGL(row_normalize_kernel, NODEG(R), x, R, xhat, inv);
    ERC_LAUNCH_CHECK("mm_row_normalize");
    return ERC_OK;
}
extern "C" int erc_mm_row_normalize_bwd(const float* xhat, const float* inv, const float* dxhat, int R, float* dx,
                                        void* stream) {
    ERC_REQUIRE(xhat && inv && dxhat && dx && R > 0, "mm_row_normalize_bwd: bad arguments");
    hipLaunchKernelGGL(row_normalize_bwd_kernel, NODEG(R), xhat, inv, dxhat, R, dx);
    ERC_LAUNCH_CHECK("mm_row_normalize_bwd");
    return ERC_OK;
}
extern "C" int erc_mm_adj_finish(const float* COS, const float* xhat, const int32_t* node_off, int B, int M, int N, int P,
                                 float* ADJ, float* CR, float* CCOS, float* DEG, void* stream) {
    ERC_REQUIRE(COS && xhat && node_off && ADJ && CR && CCOS && DEG, "mm_adj_finish: null pointer");
    ERC_REQUIRE(B > 0 && M >= 2 && M <= 3 && N > 0 && P > 0 && P <= MAXP, "mm_adj_finish: M=%d P=%d unsupported", M, P);
    const dim3 grid(erc_cdiv(M * P, 4), B);
    hipLaunchKernelGGL(adj_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, COS, xhat, node_off, M, N, P, ADJ, CCOS, DEG);
    ERC_LAUNCH_CHECK("mm_adj_rows");
    hipLaunchKernelGGL(adj_scale_kernel, grid, dim3(256), 0, (hipStream_t)stream, CCOS, DEG, node_off, M, N, P, ADJ, CR);
    ERC_LAUNCH_CHECK("mm_adj_scale");
    return ERC_OK;
}
extern "C" int erc_mm_adj_finish_bwd(const float* COS, const float* CCOS, const float* DEG, const float* dADJ,
                                     const float* dCR, const int32_t* node_off, int B, int M, int N, int P, float* G,
                                     float* GC, float* DD, void* stream) {
    ERC_REQUIRE(COS && CCOS && DEG && dADJ && dCR && node_off && G && GC && DD, "mm_adj_finish_bwd: null pointer");
    ERC_REQUIRE(B > 0 && M >= 2 && M <= 3 && N > 0 && P > 0 && P <= MAXP, "mm_adj_finish_bwd: bad sizes");
    const dim3 grid(erc_cdiv(M * P, 4), B);
    hipLaunchKernelGGL(adj_bwd_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, COS, CCOS, DEG, dADJ, dCR, node_off, M,
                       N, P, DD);
    ERC_LAUNCH_CHECK("mm_adj_bwd_rows");
    hipLaunchKernelGGL(adj_bwd_entries_kernel, grid, dim3(256), 0, (hipStream_t)stream, COS, CCOS, DEG, DD, dADJ, dCR,
                       node_off, M, N, P, G, GC);
    ERC_LAUNCH_CHECK("mm_adj_bwd_entries");
    return ERC_OK;
}
extern "C" int erc_mm_cross_apply(const float* CR, const float* h, int ldh, const int32_t* node_dlg, const int32_t* node_off,
                                  int M, int N, int P, float* out, int ldo, void* stream) {
    ERC_REQUIRE(CR && h && node_dlg && node_off && out && M >= 2 && N > 0, "mm_cross_apply: bad arguments");
    hipLaunchKernelGGL(cross_apply_kernel, NODEG(M * N), CR, h, ldh, node_dlg, node_off, M, N, P, out, ldo);
    ERC_LAUNCH_CHECK("mm_cross_apply");
    return ERC_OK;
}
extern "C" int erc_mm_cross_grad(const float* dhi, int ldd, const float* h, int ldh, const int32_t* node_dlg,
                                 const int32_t* node_off, int M, int N, int P, float* dCR, int planes, int64_t d_plane,
                                 int64_t h_plane, void* stream) {
    ERC_REQUIRE(dhi && h && node_dlg && node_off && dCR && M >= 2 && N > 0 && planes >= 1, "mm_cross_grad: bad arguments");
    hipLaunchKernelGGL(cross_grad_kernel, NODEG(M * N), dhi, ldd, h, ldh, node_dlg, node_off, M, N, P, dCR, planes, d_plane,
                       h_plane);
    ERC_LAUNCH_CHECK("mm_cross_grad");
    return ERC_OK;
}
extern "C" int erc_gcnii_combine_fwd(const float* G, const float* hi, const float* h0, int64_t n, float theta, float alpha,
                                     float drop_p, const uint64_t* rng_state, uint64_t rng_stream, float* hd,
                                     void* stream) {
    ERC_REQUIRE(G && hd && n > 0 && (!hi || h0) && (drop_p <= 0.f || rng_state), "gcnii_combine_fwd: bad arguments");
    hipLaunchKernelGGL(gcnii_combine_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, G, hi, h0, n, theta,
                       alpha, drop_p, rng_state, rng_stream, hd);
    ERC_LAUNCH_CHECK("gcnii_combine_fwd");
    return ERC_OK;
}
extern "C" int erc_gcnii_combine_bwd(const float* d_hd, const float* hd, int64_t n, float theta, float alpha,
                                     float keep_scale, int plain, float* dG, float* dhi, float* dh0, int F, int ld_d,
                                     void* stream) {
    ERC_REQUIRE(d_hd && hd && dG && n > 0 && (plain || (dhi && dh0 && F > 0 && ld_d >= F)), "gcnii_combine_bwd: bad arguments");
    hipLaunchKernelGGL(gcnii_combine_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, d_hd, hd, n, theta,
                       alpha, keep_scale, plain, dG, dhi, dh0, F > 0 ? F : 1, ld_d);
    ERC_LAUNCH_CHECK("gcnii_combine_bwd");
    return ERC_OK;
}
extern "C" int erc_dropout_fwd(const float* x, int64_t n, float drop_p, const uint64_t* rng_state, uint64_t rng_stream,
                               float* y, void* stream) {
    ERC_REQUIRE(x && y && rng_state && n > 0 && drop_p > 0.f && drop_p < 1.f, "dropout_fwd: bad arguments");
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, n, drop_p, rng_state,
                       rng_stream, y);
    ERC_LAUNCH_CHECK("dropout_fwd");
    return ERC_OK;
}
extern "C" int erc_mm_regroup_fwd(const float* xd, const float* hl, int M, int N, float drop_p, const uint64_t* rng_state,
                                  uint64_t rng_stream, float* FE, void* stream) {
    ERC_REQUIRE(xd && hl && FE && M >= 2 && N > 0 && (drop_p <= 0.f || rng_state), "mm_regroup_fwd: bad arguments");
    hipLaunchKernelGGL(regroup_fwd_kernel, dim3(ew_grid((int64_t)N * M * 2 * FD)), dim3(256), 0, (hipStream_t)stream, xd,
                       hl, M, N, drop_p, rng_state, rng_stream, FE);
    ERC_LAUNCH_CHECK("mm_regroup_fwd");
    return ERC_OK;
}
extern "C" int erc_mm_regroup_bwd(const float* dFE, const float* FE, int M, int N, float keep_scale, float* d_xd, float* d_h,
                                  void* stream) {
    ERC_REQUIRE(dFE && FE && d_xd && d_h && M >= 2 && N > 0, "mm_regroup_bwd: bad arguments");
    hipLaunchKernelGGL(regroup_bwd_kernel, dim3(ew_grid((int64_t)N * M * 2 * FD)), dim3(256), 0, (hipStream_t)stream, dFE,
                       FE, M, N, keep_scale, d_xd, d_h);
    ERC_LAUNCH_CHECK("mm_regroup_bwd");
    return ERC_OK;
}
extern "C" int erc_axpy_mask(const float* x, const float* mask, int64_t n, float scale, int accumulate, float* y,
                             void* stream) {
    ERC_REQUIRE(x && y && n > 0, "axpy_mask: bad arguments");
    hipLaunchKernelGGL(axpy_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, mask, n, scale,
                       accumulate, y);
    ERC_LAUNCH_CHECK("axpy_mask");
    return ERC_OK;
}
