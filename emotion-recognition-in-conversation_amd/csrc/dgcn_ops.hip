// DialogueGCN graph operators (track_mm/dgcn_models.py:36-152, models/rgcn.py:264-355), one wavefront per
// node, gathers over the CSRs of K1 (deterministic, no float atomics).
//
//  * EdgeAtt: per SOURCE node softmax over its window (out-edges) of (W x_k).x_j -> edge_norm
//  * basis-decomposed RGCNConv with edge_norm, add aggregation.  With W_r = sum_b att[r,b] basis[b],
//      out_i = sum_b ( sum_{e -> i} norm_e att[type_e,b] x_src(e) ) basis[b] + x_i root + bias
//    so the kernel aggregates in BASIS space, Z[i] = [30 x F], and the transform is ONE dense GEMM
//    [N, 30F] x [30F, out]: no per-edge [E, in, out] weight tensor (the reference materialises 759 MB of it,
//    models/rgcn.py:339-341) and no dependence on the number of relations (R = 162 for MELD).
//  * GraphConv's neighbour sum.
// Feature rows: F <= 256, lane l owns channels l, l+64, l+128, l+192.
#include "erc_common.h"

namespace {

constexpr int NB = 30;  // num_bases (dgcn_models.py:41)

struct Lane4 {
    float v[4];
};

__device__ __forceinline__ Lane4 load4(const float* row, int F, int lane) {
    Lane4 r;   // unconditional (clamped) loads, masked by multiplication: a guarded load is a branch + a wait
#pragma unroll
    for (int u = 0; u < 4; ++u) r.v[u] = row[min(lane + 64 * u, F - 1)] * (lane + 64 * u < F ? 1.f : 0.f);
    return r;
}
constexpr int EB = 8;   // edges per load batch: the rows of 8 edges are requested before the first is used
__device__ __forceinline__ float dot4(const Lane4& a, const Lane4& b) {
    return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2] + a.v[3] * b.v[3];
}

// dst[i,:] = src[map[i],:]  /  dst[map[i],:] = src[i,:]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int lds, const int32_t* __restrict__ map,
                                                          int N, int F, float* __restrict__ dst, int ldd, int scatter) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    const int64_t rs = scatter ? i : map[i], rd = scatter ? map[i] : i;
    for (int c = lane; c < F; c += 64) dst[rd * ldd + c] = src[rs * lds + c];
}

// ------------------------------------------------------------------ EdgeAtt
__global__ __launch_bounds__(256) void edge_att_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ att,
                                                           int lda, int F, int N, const int32_t* __restrict__ out_ptr,
                                                           const int32_t* __restrict__ out_dst,
                                                           const int32_t* __restrict__ out_eid, float* __restrict__ norm) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const Lane4 xj = load4(x + (int64_t)j * ldx, F, lane);
    const int e0 = out_ptr[j], e1 = out_ptr[j + 1];
    const int head = min(e1 - e0, 64);  // lane l keeps the score of out-edge l (window graphs: degree <= 21)
    float mx = -INFINITY, mine = -INFINITY;
    const int my_dst = out_dst[e0 + min(lane, max(head - 1, 0))];
    for (int base = 0; base < head; base += EB) {       // the first 64 out-edges, 8 attention rows in flight
        Lane4 a[EB];
#pragma unroll
        for (int u = 0; u < EB; ++u) a[u] = load4(att + (int64_t)__shfl(my_dst, min(base + u, head - 1), 64) * lda, F, lane);
#pragma unroll
        for (int u = 0; u < EB; ++u) {
            const float s = wave_sum(dot4(a[u], xj));
            if (base + u < head) mx = fmaxf(mx, s);
            if (base + u == lane) mine = s;
        }
    }
    for (int e = e0 + 64; e < e1; ++e)  // degree > 64: the maximum over the rest
        mx = fmaxf(mx, wave_sum(dot4(load4(att + (int64_t)out_dst[e] * lda, F, lane), xj)));
    const float pexp = lane < head ? expf(mine - mx) : 0.f;
    float den = wave_sum(pexp);
    for (int e = e0 + 64; e < e1; ++e)  // degree > 64: recompute
        den += expf(wave_sum(dot4(load4(att + (int64_t)out_dst[e] * lda, F, lane), xj)) - mx);
    if (lane < head) norm[out_eid[e0 + lane]] = pexp / den;
    for (int e = e0 + 64; e < e1; ++e) {
        const float s = wave_sum(dot4(load4(att + (int64_t)out_dst[e] * lda, F, lane), xj));
        if (lane == 0) norm[out_eid[e]] = expf(s - mx) / den;
    }
}

__device__ __forceinline__ void rel_sum_body(const int r, const float* __restrict__ TT, const int32_t* __restrict__ typ,
                                             const int32_t* __restrict__ counts, float* __restrict__ datt);

// what erc_edge_att_bwd_fused folds into the source-side launch (all optional)
struct EdgeBwdExtra {
    const float* dx_slabs; int n_dx_slabs; int64_t dx_slab_stride;     // dx_j += sum_s dx_slabs[s * stride + j * F + c] (the RGCN backward's partial feature gradients)
    const float* rs_TT; const int32_t* rs_typ; const int32_t* rs_counts; float* rs_datt; int rs_R;   // rs_R extra workgroups: d att[r, :] = sum_{e: type r} TT[e, :]
    int node_blocks;
};

// source side of the backward: ds_e = a_e (dnorm_e - sum a dnorm); dx_j = sum_e ds_e att_dst(e)
__global__ __launch_bounds__(256) void edge_att_bwd_source_kernel(const float* __restrict__ att, int lda, int F, int N,
                                                                  const int32_t* __restrict__ out_ptr,
                                                                  const int32_t* __restrict__ out_dst,
                                                                  const int32_t* __restrict__ out_eid,
                                                                  const float* __restrict__ norm,
                                                                  const float* __restrict__ dnorm, float* __restrict__ dx,
                                                                  int lddx, float* __restrict__ dscore, int accumulate,
                                                                  int dn_parts, int64_t dn_stride, const EdgeBwdExtra ex) {
    if ((int)blockIdx.x >= ex.node_blocks) {      // (uniform per workgroup) the relation sums of the RGCN backward ride along
        rel_sum_body((int)blockIdx.x - ex.node_blocks, ex.rs_TT, ex.rs_typ, ex.rs_counts, ex.rs_datt);
        return;
    }
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const int e0 = out_ptr[j], e1 = out_ptr[j + 1];
    // d norm may arrive as dn_parts partial vectors (the basis groups of erc_brgcn_bwd_edges_tile), summed here in order
    auto dn = [&](int id) {
        float v = dnorm[id];
        for (int s = 1; s < dn_parts; ++s) v += dnorm[(int64_t)s * dn_stride + id];
        return v;
    };
    float t = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) t += norm[out_eid[e]] * dn(out_eid[e]);
    t = wave_sum(t);
    Lane4 acc = {{0.f, 0.f, 0.f, 0.f}};
    for (int w0 = e0; w0 < e1; w0 += 64) {      // lane-parallel edge metadata, then 8 attention rows per batch
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int id = out_eid[el], my_dst = out_dst[el];
        const float my_ds = lane < nwin ? norm[id] * (dn(id) - t) : 0.f;
        if (lane < nwin) dscore[id] = my_ds;
        for (int base = 0; base < nwin; base += EB) {
            Lane4 a[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) a[u] = load4(att + (int64_t)__shfl(my_dst, min(base + u, nwin - 1), 64) * lda, F, lane);
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const float ds = __shfl(my_ds, min(base + u, 63), 64) * (base + u < nwin ? 1.f : 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc.v[q] += ds * a[u].v[q];
            }
        }
    }
    float* d = dx + (int64_t)j * lddx;
    if (ex.dx_slabs) {      // (uniform) partial feature gradients of erc_brgcn_bwd_source_tile, summed in slab order as erc_slab_reduce
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < ex.n_dx_slabs; ++sl) {
            const float* ps = ex.dx_slabs + sl * ex.dx_slab_stride + (int64_t)j * F;
#pragma unroll
            for (int u = 0; u < 4; ++u) part[u] += ps[min(lane + 64 * u, F - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc.v[u] += part[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < F) d[lane + 64 * u] = accumulate ? d[lane + 64 * u] + acc.v[u] : acc.v[u];
}

// target side: datt_k = sum_{e into k} dscore_e x_src(e)
__global__ __launch_bounds__(256) void edge_att_bwd_target_kernel(const float* __restrict__ x, int ldx, int F, int N,
                                                                  const int32_t* __restrict__ in_ptr,
                                                                  const int32_t* __restrict__ in_src,
                                                                  const float* __restrict__ dscore,
                                                                  float* __restrict__ datt, int ldda) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= N) return;
    Lane4 acc = {{0.f, 0.f, 0.f, 0.f}};
    const int e0 = in_ptr[k], e1 = in_ptr[k + 1];
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_src = in_src[el];
        const float my_ds = lane < nwin ? dscore[el] : 0.f;
        for (int base = 0; base < nwin; base += EB) {
            Lane4 xs[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) xs[u] = load4(x + (int64_t)__shfl(my_src, min(base + u, nwin - 1), 64) * ldx, F, lane);
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const float ds = __shfl(my_ds, min(base + u, 63), 64) * (base + u < nwin ? 1.f : 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc.v[q] += ds * xs[u].v[q];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < F) datt[(int64_t)k * ldda + lane + 64 * u] = acc.v[u];
}

// ------------------------------------------------------------------ basis RGCN
// Z[i, b*F + c] = sum_{e into i} norm_e att[type_e, b] x[src_e, c]
__global__ __launch_bounds__(256) void brgcn_agg_fwd_kernel(const float* __restrict__ x, int ldx, int F, int N,
                                                            const int32_t* __restrict__ in_ptr,
                                                            const int32_t* __restrict__ in_src,
                                                            const int32_t* __restrict__ in_typ,
                                                            const float* __restrict__ norm, const float* __restrict__ attw,
                                                            float* __restrict__ Z) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    float acc[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[b][u] = 0.f;
    for (int e = in_ptr[i]; e < in_ptr[i + 1]; ++e) {
        const Lane4 xs = load4(x + (int64_t)in_src[e] * ldx, F, lane);
        const float ne = norm[e];
        const float* ar = attw + (int64_t)in_typ[e] * NB;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float c = ne * ar[b];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[b][u] += c * xs.v[u];
        }
    }
    float* z = Z + (int64_t)i * NB * F;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < F) z[b * F + lane + 64 * u] = acc[b][u];
}

// Sum 16 per-lane values over the wavefront in 17 shuffles (instead of 16 x 6): halve the set of values and the lane
// distance together.  Lane l ends up with the total of entry ((l>>5)&1)*8 + ((l>>4)&1)*4 + ((l>>3)&1)*2 + ((l>>2)&1).
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
    return e;
}

// per target i, per in-edge e: T_e[b] = x_src . dZ[i, b, :]  ->  dnorm_e = sum_b att[type_e,b] T_e[b],
// TT[e, b] = norm_e T_e[b]  (summed per relation by rel_sum_kernel -> d att).
// The 30 dot products of an edge are reduced by two 16-value butterflies (34 shuffles; one full wave_sum per basis was
// 180), and the source rows of 4 edges are requested together.
__global__ __launch_bounds__(256) void brgcn_bwd_target_kernel(const float* __restrict__ x, int ldx, int F, int N,
                                                               const int32_t* __restrict__ in_ptr,
                                                               const int32_t* __restrict__ in_src,
                                                               const int32_t* __restrict__ in_typ,
                                                               const float* __restrict__ norm, const float* __restrict__ attw,
                                                               const float* __restrict__ dZ, float* __restrict__ dnorm,
                                                               float* __restrict__ TT) {
    static_assert(NB <= 32, "two butterflies of 16");
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    float dz[NB][4];
    const float* z = dZ + (int64_t)i * NB * F;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < 4; ++u) dz[b][u] = z[b * F + min(lane + 64 * u, F - 1)] * (lane + 64 * u < F ? 1.f : 0.f);
    const int entry = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    const bool rep = (lane & 3) == 0;          // one of the 4 lanes that hold the same entry
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    for (int eb = e0; eb < e1; eb += 4) {
        Lane4 xs[4];
        int typ[4];
        float nrm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(eb + u, e1 - 1);
            const float* row = x + (int64_t)in_src[e] * ldx;
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[u].v[q] = row[min(lane + 64 * q, F - 1)] * (lane + 64 * q < F ? 1.f : 0.f);
            typ[u] = in_typ[e];
            nrm[u] = norm[e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (eb + u < e1) {     // uniform
                const int e = eb + u;
                float pa[16], pb[16];
#pragma unroll
                for (int b = 0; b < 16; ++b)
                    pa[b] = xs[u].v[0] * dz[b][0] + xs[u].v[1] * dz[b][1] + xs[u].v[2] * dz[b][2] + xs[u].v[3] * dz[b][3];
#pragma unroll
                for (int b = 0; b < 16; ++b)
                    pb[b] = b + 16 < NB ? xs[u].v[0] * dz[(b + 16) % NB][0] + xs[u].v[1] * dz[(b + 16) % NB][1] +
                                              xs[u].v[2] * dz[(b + 16) % NB][2] + xs[u].v[3] * dz[(b + 16) % NB][3]
                                        : 0.f;
                const float ta = butterfly16_sum(pa, lane);      // T_e[entry]
                const float tb = butterfly16_sum(pb, lane);      // T_e[16 + entry]
                const bool vb = entry + 16 < NB;
                const float* aw = attw + (int64_t)typ[u] * NB;
                const float wa = aw[entry], wb = aw[min(entry + 16, NB - 1)];
                const float dn = wave_sum(rep ? wa * ta + (vb ? wb * tb : 0.f) : 0.f);
                if (lane == 0) dnorm[e] = dn;
                if (rep) {
                    TT[(int64_t)e * NB + entry] = nrm[u] * ta;
                    if (vb) TT[(int64_t)e * NB + entry + 16] = nrm[u] * tb;
                }
            }
        }
    }
}

// datt[r, b] = sum_{e: type_e == r} TT[e, b]; one workgroup per relation, fixed summation order.
// Phase 1: every wavefront scans a contiguous quarter of the edge types (coalesced, 8 loads in flight) and appends the
// matching edge ids to its own LDS list with ballot / popcount -- no barrier, lists stay in edge order.  Phase 2: the
// four lists are walked by 8 slots x 32 lanes (8 rows of TT in flight per slot); slots are combined in fixed order.
// (The first version tested typ[e] == r per edge inside the accumulation loop: ~1750 dependent guarded loads per
// thread, 120 us for E = 14 k edges; this one 1/10 of that.)
constexpr int RS_CAP = 2048;  // matching edges kept per wavefront; more fall back to the scan-and-add path

__device__ __forceinline__ void rel_sum_body(const int r, const float* __restrict__ TT, const int32_t* __restrict__ typ,
                                             const int32_t* __restrict__ counts, float* __restrict__ datt) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int E = counts[1];
    if (E <= 0) {
        if (tid < NB) datt[r * NB + tid] = 0.f;
        return;
    }
    __shared__ int list[4][RS_CAP];
    __shared__ int cnt[4];
    __shared__ float sh[8][32];
    const int per = (E + 3) / 4, e_lo = w * per, e_hi = min(E, e_lo + per);
    int n = 0;
    for (int e0 = e_lo; e0 < e_hi; e0 += 8 * 64) {
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = typ[min(e0 + u * 64 + lane, E - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 64 + lane;
            const bool hit = e < e_hi && t[u] == r;
            const unsigned long long m = __ballot(hit);
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && pos < RS_CAP) list[w][pos] = e;
            n += __popcll(m);
        }
    }
    if (lane == 0) cnt[w] = n;
    __syncthreads();
    const int b = tid & 31, slot = tid >> 5;  // 8 slots x 32 lanes (NB = 30 used)
    const int bc = min(b, NB - 1);
    float acc = 0.f;
    const bool overflow = cnt[0] > RS_CAP || cnt[1] > RS_CAP || cnt[2] > RS_CAP || cnt[3] > RS_CAP;
    if (!overflow) {
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const int c = cnt[ww];
            for (int i0 = slot; i0 < c; i0 += 8 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = TT[(int64_t)list[ww][min(i0 + 8 * u, c - 1)] * NB + bc];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u] * (i0 + 8 * u < c ? 1.f : 0.f);
            }
        }
    } else {
        for (int e = slot; e < E; e += 8)
            if (typ[e] == r) acc += TT[(int64_t)e * NB + bc];
    }
    sh[slot][b] = acc;
    __syncthreads();
    if (slot == 0 && b < NB) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += sh[k][b];
        datt[r * NB + b] = s;
    }
}

__global__ __launch_bounds__(256) void rel_sum_kernel(const float* __restrict__ TT, const int32_t* __restrict__ typ,
                                                      const int32_t* __restrict__ counts, float* __restrict__ datt) {
    rel_sum_body(blockIdx.x, TT, typ, counts, datt);
}

// U[j, b*O + c] = sum_{e out of j} norm_e att[type_e, b] dH[dst_e, c]   (O <= 128 output channels)
__global__ __launch_bounds__(256) void brgcn_bwd_source_kernel(const float* __restrict__ dH, int lddh, int O, int N,
                                                               const int32_t* __restrict__ out_ptr,
                                                               const int32_t* __restrict__ out_dst,
                                                               const int32_t* __restrict__ out_typ,
                                                               const int32_t* __restrict__ out_eid,
                                                               const float* __restrict__ norm, const float* __restrict__ attw,
                                                               float* __restrict__ U) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    float acc[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b][0] = acc[b][1] = 0.f;
    const bool h0 = lane < O, h1 = lane + 64 < O;
    for (int e = out_ptr[j]; e < out_ptr[j + 1]; ++e) {
        const float* g = dH + (int64_t)out_dst[e] * lddh;
        const float g0 = h0 ? g[lane] : 0.f, g1 = h1 ? g[lane + 64] : 0.f;
        const float ne = norm[out_eid[e]];
        const float* ar = attw + (int64_t)out_typ[e] * NB;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float c = ne * ar[b];
            acc[b][0] += c * g0;
            acc[b][1] += c * g1;
        }
    }
    float* u = U + (int64_t)j * NB * O;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (h0) u[b * O + lane] = acc[b][0];
        if (h1) u[b * O + lane + 64] = acc[b][1];
    }
}

// ------------------------------------------------------------------ basis RGCN forward as ONE tile launch
// conv(x) = Z @ basis.view(30F, O) + x @ root + bias with F = 200, O = 100 was four launches: the aggregate (Z = 24 KB per
// node to HBM), a split-K GEMM with K = 6000 on 16 x 32 wave tiles (4-byte weight-fragment loads), the root GEMM, the slab
// reduce -- 72 us at N = 700.  Here a workgroup owns 16 nodes x 5 bases (blockIdx.y = basis group; the last group also
// carries the root term as a 6th block, Z_root = x): it aggregates its Z blocks into LDS (and to HBM for the weight
// gradient), multiplies them by its 1000 (1200) rows of basis on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32),
// K split over the 8 wavefronts, and leaves a [16, 100] partial in slab blockIdx.y; erc_slab_reduce adds the slabs + bias.
// (The product is matrix-core bound -- 16 x 128 x 6200 MACs per tile at 256 per cycle and CU -- so the groups are sized to
// put the layer on the whole chip: 10-basis groups used 132 CUs and took 20 us for it, the aggregate another 23.)
// Weight fragments are 16-byte loads: a lane's float4 = 4 neighbouring output columns of ONE k, used by 4 MFMAs whose
// tiles interleave the columns (column 64 h + 4 n + j belongs to MFMA (h, j), lane column n).
constexpr int TF = 200, TO = 100, TG = 5, NGRP = NB / TG;      // features, outputs, bases per group, groups
constexpr int TKP = (TG + 1) * TF + 4;          // LDS row of the Z tile (10 blocks + root block), 16-byte multiple
typedef float f32x4_t __attribute__((ext_vector_type(4)));
unsigned long long* g_brgcn_stamps = nullptr;      // diagnostic: 100 MHz stamps of one workgroup (erc_brgcn_set_stamps)
#define BR_STAMP(slot)                                                                                              \
    do {                                                                                                            \
        if (stamps && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y - 1 && threadIdx.x == 0)                           \
            stamps[slot] = __builtin_amdgcn_s_memrealtime();                                                        \
    } while (0)

__global__ __launch_bounds__(512) void brgcn_fwd_tile_kernel(const float* __restrict__ x, int ldx, int N,
                                                             const int32_t* __restrict__ in_ptr,
                                                             const int32_t* __restrict__ in_src,
                                                             const int32_t* __restrict__ in_typ,
                                                             const float* __restrict__ norm, const float* __restrict__ attw,
                                                             const float* __restrict__ basis, const float* __restrict__ root,
                                                             float* __restrict__ Z, float* __restrict__ slabs,
                                                             unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BR_STAMP(0);
    float* Zt = smem;                                    // [16][TKP]; later the 8 partial tiles [8][16][128]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * 16, g = blockIdx.y;
    const bool with_root = g == NGRP - 1;
    // ---- phase 1: Z[i, b, :] = sum_{e into i} norm_e att[type_e, b] x[src_e, :] for this group's bases.  A wavefront
    // advances its two nodes together (8 source rows in flight); an edge's 5 coefficients norm_e att[type_e, b] are computed
    // by the lane that holds the edge's metadata and read with v_readlane: no memory operation per edge but the source row.
    {
        const int liA = 2 * w, liB = 2 * w + 1, iA = i0 + liA, iB = i0 + liB;
        float accA[TG][4], accB[TG][4];
#pragma unroll
        for (int b = 0; b < TG; ++b)
#pragma unroll
            for (int u = 0; u < 4; ++u) accA[b][u] = accB[b][u] = 0.f;
        int eA0 = 0, eA1 = 0, eB0 = 0, eB1 = 0;
        if (iA < N) eA0 = in_ptr[iA], eA1 = in_ptr[iA + 1];
        if (iB < N) eB0 = in_ptr[iB], eB1 = in_ptr[iB + 1];
        const int dmax = max(eA1 - eA0, eB1 - eB0);
        auto rl = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
        for (int wo = 0; wo < dmax; wo += 64) {
            const int nA = min(64, max(eA1 - eA0 - wo, 0)), nB = min(64, max(eB1 - eB0 - wo, 0));
            const int elA = nA > 0 ? eA0 + wo + min(lane, nA - 1) : 0, elB = nB > 0 ? eB0 + wo + min(lane, nB - 1) : 0;
            const int srcA = in_src[elA], srcB = in_src[elB];
            const float nrA = lane < nA ? norm[elA] : 0.f, nrB = lane < nB ? norm[elB] : 0.f;
            const float* arA = attw + (int64_t)in_typ[elA] * NB + g * TG;
            const float* arB = attw + (int64_t)in_typ[elB] * NB + g * TG;
            float cfA[TG], cfB[TG];
#pragma unroll
            for (int b = 0; b < TG; ++b) cfA[b] = nrA * arA[b], cfB[b] = nrB * arB[b];
            const int nmax = max(nA, nB);
            for (int base = 0; base < nmax; base += 4) {
                Lane4 xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ja = __builtin_amdgcn_readlane(srcA, min(base + u, max(nA - 1, 0)));
                    const int jb = __builtin_amdgcn_readlane(srcB, min(base + u, max(nB - 1, 0)));
                    xa[u] = load4(x + (int64_t)ja * ldx, TF, lane);
                    xb[u] = load4(x + (int64_t)jb * ldx, TF, lane);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int l = min(base + u, 63);       // lanes past a node's window hold coefficient 0
#pragma unroll
                    for (int b = 0; b < TG; ++b) {
                        const float ca = rl(cfA[b], l), cb = rl(cfB[b], l);
#pragma unroll
                        for (int q = 0; q < 4; ++q) accA[b][q] += ca * xa[u].v[q], accB[b][q] += cb * xb[u].v[q];
                    }
                }
            }
        }
        Lane4 selfA = {{0.f, 0.f, 0.f, 0.f}}, selfB = {{0.f, 0.f, 0.f, 0.f}};
        if (with_root && iA < N) selfA = load4(x + (int64_t)iA * ldx, TF, lane);
        if (with_root && iB < N) selfB = load4(x + (int64_t)iB * ldx, TF, lane);
        float* zrA = Zt + liA * TKP;
        float* zrB = Zt + liB * TKP;
        float* zgA = Z + (int64_t)min(iA, N - 1) * NB * TF + g * TG * TF;
        float* zgB = Z + (int64_t)min(iB, N - 1) * NB * TF + g * TG * TF;
#pragma unroll
        for (int b = 0; b < TG; ++b)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (lane + 64 * u < TF) {
                    zrA[b * TF + lane + 64 * u] = accA[b][u];
                    zrB[b * TF + lane + 64 * u] = accB[b][u];
                    if (iA < N) zgA[b * TF + lane + 64 * u] = accA[b][u];
                    if (iB < N) zgB[b * TF + lane + 64 * u] = accB[b][u];
                }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < TF) zrA[TG * TF + lane + 64 * u] = selfA.v[u], zrB[TG * TF + lane + 64 * u] = selfB.v[u];
    }
    BR_STAMP(1);
    __syncthreads();
    BR_STAMP(2);
    // ---- phase 2: partial[16, 100] = Zt[16, K] @ rows of [basis group ; root], K split over the wavefronts
    const int nks = (with_root ? (TG + 1) * TF : TG * TF) / 4;          // k-steps of 4
    const int per = (nks + 7) / 8, ks0 = w * per, ks1 = min(nks, ks0 + per);
    const int r = lane & 15, kk = lane >> 4;
    const float* bg = basis + (int64_t)g * TG * TF * TO;
    f32x4_t acc[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int c0 = 4 * r, c1 = 64 + 4 * r;              // the lane's 4 columns in the two column halves
    const bool v1 = c1 < TO;                             // 64 + 4 r + 3 < 100  <=>  r < 9 (whole float4 inside)
    const int c1c = v1 ? c1 : 0;
    constexpr int PB = 8;      // k-steps per batch; two batches in flight (the stream is latency-bound: 63 k-steps per wavefront)
    struct Batch { float4 b0[PB], b1[PB]; float a[PB]; };
    auto fetch = [&](int ks, Batch& t) {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int k = 4 * min(ks + u, ks1 - 1) + kk;
            const float* brow = k < TG * TF ? bg + (int64_t)k * TO : root + (int64_t)(k - TG * TF) * TO;
            t.b0[u] = *reinterpret_cast<const float4*>(brow + c0);
            t.b1[u] = *reinterpret_cast<const float4*>(brow + c1c);
            t.a[u] = Zt[r * TKP + k];
        }
    };
    const float m1 = v1 ? 1.f : 0.f;
    auto consume = [&](int ks, const Batch& t) {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const float av = ks + u < ks1 ? t.a[u] : 0.f;
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t.b0[u].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t.b0[u].y, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t.b0[u].z, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t.b0[u].w, acc[0][3], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t.b1[u].x * m1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t.b1[u].y * m1, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t.b1[u].z * m1, acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, t.b1[u].w * m1, acc[1][3], 0, 0, 0);
        }
    };
    Batch tA, tB;
    if (ks0 < ks1) fetch(ks0, tA);
    for (int ks = ks0; ks < ks1; ks += 2 * PB) {
        if (ks + PB < ks1) fetch(ks + PB, tB);
        consume(ks, tA);
        if (ks + PB < ks1) {
            if (ks + 2 * PB < ks1) fetch(ks + 2 * PB, tA);
            consume(ks + PB, tB);
        }
    }
    BR_STAMP(3);
    __syncthreads();            // every wavefront is done with the Z tile: its LDS becomes the partial tiles
    BR_STAMP(4);
    // D of MFMA (h, j): lane holds rows 4 (lane >> 4) + i, i < 4, of tile column n = lane & 15 = output column 64 h + 4 n + j
    float* red = smem + w * 16 * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(red + (4 * kk + i) * 128 + 64 * h + 4 * r) =
                float4{acc[h][0][i], acc[h][1][i], acc[h][2][i], acc[h][3][i]};
    __syncthreads();
    // ---- phase 3: sum of the 8 partial tiles -> slab g (16 rows x 25 float4)
    if (tid < 16 * 25) {
        const int row = tid / 25, c4 = tid % 25;
        float4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            const float4 v = *reinterpret_cast<const float4*>(smem + (ww * 16 + row) * 128 + 4 * c4);
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
        }
        if (i0 + row < N)
            *reinterpret_cast<float4*>(slabs + ((int64_t)g * N + i0 + row) * TO + 4 * c4) = sum;
    }
    BR_STAMP(5);
}

// The node side of the backward the same way: dx[j, :] = sum_b U[j, b, :] basis[b]^T + dOut[j, :] root^T with
// U[j, b, :] = sum_{e out of j} norm_e att[type_e, b] dOut[dst_e, :].  K = (b, c) runs along the CONTIGUOUS index of
// basis [30, F, O], so a lane's float4 is 4 consecutive k of one output column f: the matrix core sums over k whatever
// slot a k sits in, so MFMA step j of a 16-k block takes element j of the float4 on both operands (A rows from LDS the
// same way) and no transposed copy of basis is needed.
constexpr int SKB = TG * TO, SKR = SKB + TO;          // k of a group (500), with the root block (600)
constexpr int SKP = 608 + 4;                           // LDS row: K padded to a multiple of 16 (zeros) + 4
constexpr int SNT = 13;                                // 16-column tiles over F = 200 (208)

__global__ __launch_bounds__(512) void brgcn_bwd_source_tile_kernel(const float* __restrict__ dH, int lddh, int N,
                                                                    const int32_t* __restrict__ out_ptr,
                                                                    const int32_t* __restrict__ out_dst,
                                                                    const int32_t* __restrict__ out_typ,
                                                                    const int32_t* __restrict__ out_eid,
                                                                    const float* __restrict__ norm,
                                                                    const float* __restrict__ attw,
                                                                    const float* __restrict__ basis,
                                                                    const float* __restrict__ root, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ut = smem;                                    // [16][SKP]; later the 8 partial tiles [8][16][208]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * 16, g = blockIdx.y;
    const bool with_root = g == NGRP - 1;
    const int c0 = min(lane, TO - 1), c1 = min(lane + 64, TO - 1);
    const float m0 = lane < TO ? 1.f : 0.f, m1 = lane + 64 < TO ? 1.f : 0.f;
    // ---- phase 1: the U blocks of this group, two nodes per wavefront advanced together
    {
        const int liA = 2 * w, liB = 2 * w + 1, jA = j0 + liA, jB = j0 + liB;
        float accA[TG][2], accB[TG][2];
#pragma unroll
        for (int b = 0; b < TG; ++b) accA[b][0] = accA[b][1] = accB[b][0] = accB[b][1] = 0.f;
        int eA0 = 0, eA1 = 0, eB0 = 0, eB1 = 0;
        if (jA < N) eA0 = out_ptr[jA], eA1 = out_ptr[jA + 1];
        if (jB < N) eB0 = out_ptr[jB], eB1 = out_ptr[jB + 1];
        const int dmax = max(eA1 - eA0, eB1 - eB0);
        auto rl = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
        for (int wo = 0; wo < dmax; wo += 64) {
            const int nA = min(64, max(eA1 - eA0 - wo, 0)), nB = min(64, max(eB1 - eB0 - wo, 0));
            const int elA = nA > 0 ? eA0 + wo + min(lane, nA - 1) : 0, elB = nB > 0 ? eB0 + wo + min(lane, nB - 1) : 0;
            const int dstA = out_dst[elA], dstB = out_dst[elB];
            const float nrA = lane < nA ? norm[out_eid[elA]] : 0.f, nrB = lane < nB ? norm[out_eid[elB]] : 0.f;
            const float* arA = attw + (int64_t)out_typ[elA] * NB + g * TG;
            const float* arB = attw + (int64_t)out_typ[elB] * NB + g * TG;
            float cfA[TG], cfB[TG];
#pragma unroll
            for (int b = 0; b < TG; ++b) cfA[b] = nrA * arA[b], cfB[b] = nrB * arB[b];
            const int nmax = max(nA, nB);
            for (int base = 0; base < nmax; base += 4) {
                float ga[4][2], gb[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* ra = dH + (int64_t)__builtin_amdgcn_readlane(dstA, min(base + u, max(nA - 1, 0))) * lddh;
                    const float* rb = dH + (int64_t)__builtin_amdgcn_readlane(dstB, min(base + u, max(nB - 1, 0))) * lddh;
                    ga[u][0] = ra[c0] * m0, ga[u][1] = ra[c1] * m1;
                    gb[u][0] = rb[c0] * m0, gb[u][1] = rb[c1] * m1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int l = min(base + u, 63);       // lanes past a node's window hold coefficient 0
#pragma unroll
                    for (int b = 0; b < TG; ++b) {
                        const float ca = rl(cfA[b], l), cb = rl(cfB[b], l);
                        accA[b][0] += ca * ga[u][0], accA[b][1] += ca * ga[u][1];
                        accB[b][0] += cb * gb[u][0], accB[b][1] += cb * gb[u][1];
                    }
                }
            }
        }
        float* urA = Ut + liA * SKP;
        float* urB = Ut + liB * SKP;
#pragma unroll
        for (int b = 0; b < TG; ++b) {
            if (lane < TO) urA[b * TO + lane] = accA[b][0], urB[b * TO + lane] = accB[b][0];
            if (lane + 64 < TO) urA[b * TO + lane + 64] = accA[b][1], urB[b * TO + lane + 64] = accB[b][1];
        }
        // root block (last group: the node's own dOut row, else zeros) and the zero padding up to 608
        const float* sa = dH + (int64_t)min(jA, N - 1) * lddh;
        const float* sb = dH + (int64_t)min(jB, N - 1) * lddh;
        const float ka = with_root && jA < N ? 1.f : 0.f, kb = with_root && jB < N ? 1.f : 0.f;
        if (lane < TO) urA[SKB + lane] = sa[c0] * ka, urB[SKB + lane] = sb[c0] * kb;
        if (lane + 64 < TO) urA[SKB + lane + 64] = sa[c1] * ka, urB[SKB + lane + 64] = sb[c1] * kb;
        if (lane < SKP - SKR) urA[SKR + lane] = 0.f, urB[SKR + lane] = 0.f;
    }
    __syncthreads();
    // ---- phase 2: partial[16, 200] = Ut[16, K] @ B, B[k = (b, c)][f] = basis[5 g + b][f][c] (root[f][c] for the 6th block);
    //      16-k blocks dealt to the wavefronts round-robin
    const int nblk = (with_root ? SKR + 8 : SKB + 12) / 16;      // 600 -> 38 blocks (608), 500 -> 32 blocks (512)
    const int r = lane & 15, kk = lane >> 4;
    const float* bg = basis + (int64_t)g * TG * TF * TO;
    f32x4_t acc[SNT];
#pragma unroll
    for (int t = 0; t < SNT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int blk = w; blk < nblk; blk += 8) {
        const int k = 16 * blk + 4 * kk;                 // this lane's 4 consecutive k (one basis: 4 | 100)
        const int kc = min(k, (with_root ? SKR : SKB) - 4);
        const bool kv = k < (with_root ? SKR : SKB);
        const float* bcol = kc < SKB ? bg + (int64_t)(kc / TO) * TF * TO + kc % TO : root + (kc - SKB);
        const float4 av = *reinterpret_cast<const float4*>(Ut + r * SKP + k);      // zeros past K
        float4 bv[SNT];
#pragma unroll
        for (int t = 0; t < SNT; ++t) {
            const int f = min(16 * t + r, TF - 1);
            bv[t] = *reinterpret_cast<const float4*>(bcol + (int64_t)f * TO);
        }
#pragma unroll
        for (int t = 0; t < SNT; ++t) {
            const float mk = kv && 16 * t + r < TF ? 1.f : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[t].x * mk, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[t].y * mk, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[t].z * mk, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[t].w * mk, acc[t], 0, 0, 0);
        }
    }
    __syncthreads();
    // D of tile t: lane holds rows 4 (lane >> 4) + i of column 16 t + (lane & 15)
    float* red = smem + w * 16 * 208;
#pragma unroll
    for (int t = 0; t < SNT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(4 * kk + i) * 208 + 16 * t + r] = acc[t][i];
    __syncthreads();
    // ---- phase 3: sum of the 8 partial tiles -> slab g (16 rows x 50 float4)
    for (int it = tid; it < 16 * 50; it += 512) {
        const int row = it / 50, c4 = it % 50;
        float4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            const float4 v = *reinterpret_cast<const float4*>(smem + (ww * 16 + row) * 208 + 4 * c4);
            sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
        }
        if (j0 + row < N) *reinterpret_cast<float4*>(slabs + ((int64_t)g * N + j0 + row) * TF + 4 * c4) = sum;
    }
}

// The edge side of the backward: dZ[i, b, :] = basis[b] dOut[i, :] never goes to HBM.  Per 16 target nodes x 5 bases the
// workgroup forms its dZ blocks on the matrix cores (K = the 100 outputs, 16-byte weight loads along k as above) into LDS,
// then every in-edge takes 6 dot products with its source row -- T_e[b] = x_src . dZ[i, b, :] for the group's bases and
// x_src . (sum_b att[type_e, b] dZ[i, b, :]) = this group's share of d norm_e -- two edges per 16-value butterfly.
// Outputs: TT[e, 5 g + b] = norm_e T_e[b] (summed per relation by rel_sum_kernel -> d att) and slab g of d norm.
constexpr int DZP = TG * TF + 4;          // LDS row of the dZ tile
constexpr int DKB = 7;                    // 16-k blocks over the 100 outputs (112, masked)

__global__ __launch_bounds__(512) void brgcn_bwd_target_tile_kernel(const float* __restrict__ x, int ldx, int N,
                                                                    const int32_t* __restrict__ in_ptr,
                                                                    const int32_t* __restrict__ in_src,
                                                                    const int32_t* __restrict__ in_typ,
                                                                    const float* __restrict__ norm,
                                                                    const float* __restrict__ attw,
                                                                    const float* __restrict__ basis,
                                                                    const float* __restrict__ dH, int lddh,
                                                                    float* __restrict__ TT, float* __restrict__ dn_slabs,
                                                                    int64_t dn_stride, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BR_STAMP(0);
    float* dZt = smem;                                   // [16][DZP]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * 16, g = blockIdx.y;
    const int r = lane & 15, kk = lane >> 4;
    // ---- phase A: dZ tile [16, 5 x 200] = dOut[16, 100] @ B, B[k = c][n = (b, f)] = basis[5 g + b][f][c]
    {
        float4 av[DKB];
        const float* arow = dH + (int64_t)min(i0 + r, N - 1) * lddh;
        const float rowok = i0 + r < N ? 1.f : 0.f;
#pragma unroll
        for (int blk = 0; blk < DKB; ++blk) {
            const int k = 16 * blk + 4 * kk;
            const float4 v = *reinterpret_cast<const float4*>(arow + min(k, TO - 4));
            const float m = k < TO ? rowok : 0.f;
            av[blk] = float4{v.x * m, v.y * m, v.z * m, v.w * m};
        }
        const float* bg = basis + (int64_t)g * TG * TF * TO;
        constexpr int NTL = (TG * TF + 15) / 16;        // 63 column tiles (the last one half)
        auto fetch_b = [&](int t, float4 (&bv)[DKB]) {
            const int col = min(16 * min(t, NTL - 1) + r, TG * TF - 1);
            const float* bcol = bg + (int64_t)col * TO;   // (b, f) -> basis[5 g + b][f][:], rows of 100 are consecutive
#pragma unroll
            for (int blk = 0; blk < DKB; ++blk) bv[blk] = *reinterpret_cast<const float4*>(bcol + min(16 * blk + 4 * kk, TO - 4));
        };
        auto tile = [&](int t, const float4 (&bv)[DKB]) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < DKB; ++blk) {       // A is zero where k >= 100: the clamped B values do not matter
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[blk].x, bv[blk].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[blk].y, bv[blk].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[blk].z, bv[blk].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[blk].w, bv[blk].w, acc, 0, 0, 0);
            }
            if (16 * t + r < TG * TF) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dZt[(4 * kk + i) * DZP + 16 * t + r] = acc[i];
            }
        };
        // the weight fragments of the next tile are requested before the current tile's products (two register sets)
        // (every request is consumed: a build that left clamped dummy requests in flight at the end of the loop was not
        // bit-reproducible, DESIGN.md finding 30)
        float4 bA[DKB], bB[DKB];
        fetch_b(w, bA);
        for (int t = w; t < NTL; t += 16) {
            if (t + 8 < NTL) fetch_b(t + 8, bB);
            tile(t, bA);
            if (t + 8 < NTL) {
                if (t + 16 < NTL) fetch_b(t + 16, bA);
                tile(t + 8, bB);
            }
        }
    }
    BR_STAMP(1);
    __syncthreads();
    BR_STAMP(2);
    // ---- phase B: per in-edge of the tile's nodes, the 6 dot products with the source row
    const int entry = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    const int eu = entry / 6, ev = entry % 6;            // entries 0..11: edge eu of the pair, value ev (5 = d norm share)
    auto rl = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    {
        const int liA = 2 * w, liB = 2 * w + 1, iA = i0 + liA, iB = i0 + liB;
        Lane4 dzA[TG], dzB[TG];
#pragma unroll
        for (int b = 0; b < TG; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                dzA[b].v[q] = lane + 64 * q < TF ? dZt[liA * DZP + b * TF + lane + 64 * q] : 0.f;
                dzB[b].v[q] = lane + 64 * q < TF ? dZt[liB * DZP + b * TF + lane + 64 * q] : 0.f;
            }
        int eA0 = 0, eA1 = 0, eB0 = 0, eB1 = 0;
        if (iA < N) eA0 = in_ptr[iA], eA1 = in_ptr[iA + 1];
        if (iB < N) eB0 = in_ptr[iB], eB1 = in_ptr[iB + 1];
        const int dmax = max(eA1 - eA0, eB1 - eB0);
        // one butterfly = one edge of each node: entries 0..5 node A's edge, 6..11 node B's
        for (int wo = 0; wo < dmax; wo += 64) {
            const int nA = min(64, max(eA1 - eA0 - wo, 0)), nB = min(64, max(eB1 - eB0 - wo, 0));
            const int elA = nA > 0 ? eA0 + wo + min(lane, nA - 1) : 0, elB = nB > 0 ? eB0 + wo + min(lane, nB - 1) : 0;
            const int srcA = in_src[elA], srcB = in_src[elB];
            const float nrA = lane < nA ? norm[elA] : 0.f, nrB = lane < nB ? norm[elB] : 0.f;
            const float* arA = attw + (int64_t)in_typ[elA] * NB + g * TG;
            const float* arB = attw + (int64_t)in_typ[elB] * NB + g * TG;
            float atA[TG], atB[TG];
#pragma unroll
            for (int b = 0; b < TG; ++b) atA[b] = arA[b], atB[b] = arB[b];
            const int nmax = max(nA, nB);
            for (int base = 0; base < nmax; base += 4) {      // 4 + 4 source rows in flight
                Lane4 xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xa[u] = load4(x + (int64_t)__builtin_amdgcn_readlane(srcA, min(base + u, max(nA - 1, 0))) * ldx, TF, lane);
                    xb[u] = load4(x + (int64_t)__builtin_amdgcn_readlane(srcB, min(base + u, max(nB - 1, 0))) * ldx, TF, lane);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (base + u < nmax) {      // uniform
                        const int l = min(base + u, 63);
                        float part[16];
                        Lane4 cA = {{0.f, 0.f, 0.f, 0.f}}, cB = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                        for (int b = 0; b < TG; ++b) {
                            part[b] = dot4(xa[u], dzA[b]);
                            part[6 + b] = dot4(xb[u], dzB[b]);
                            const float ca = rl(atA[b], l), cb = rl(atB[b], l);
#pragma unroll
                            for (int q = 0; q < 4; ++q) cA.v[q] += ca * dzA[b].v[q], cB.v[q] += cb * dzB[b].v[q];
                        }
                        part[5] = dot4(xa[u], cA);
                        part[11] = dot4(xb[u], cB);
                        part[12] = part[13] = part[14] = part[15] = 0.f;
                        const float tot = butterfly16_sum(part, lane);
                        const float n0 = rl(nrA, l), n1 = rl(nrB, l);
                        const bool ok = eu == 0 ? base + u < nA : base + u < nB;
                        if ((lane & 3) == 0 && entry < 12 && ok) {
                            const int64_t e = (eu == 0 ? eA0 : eB0) + wo + base + u;
                            if (ev < TG) TT[e * NB + g * TG + ev] = (eu ? n1 : n0) * tot;
                            else dn_slabs[(int64_t)g * dn_stride + e] = tot;
                        }
                    }
                }
            }
        }
        BR_STAMP(3);
    }
    BR_STAMP(4);
}

// ------------------------------------------------------------------ basis RGCN in RELATION space (R <= 8)
// With few relations (two speakers: R = 2 S^2 = 8 < 30 bases) the layer is cheaper the way models/rgcn.py:300-304 writes
// it: W_r = sum_b comp[r,b] basis[b] first, then
//      out_i = sum_r ( sum_{e -> i, type r} norm_e x_src(e) ) W_r + x_i root + bias
// -- Z is [N, R F] instead of [N, 30 F] (the two GEMMs, the weight gradient and the scatter shrink by 30/R) and the
// per-edge work is one multiply-add row instead of 30.  comp / basis receive their gradients from dW_r afterwards.
constexpr int RR = 8;

// Wr[r][k] = sum_b comp[r,b] basis[b][k] (k < F*O, row-major [F,O]); WrT[r][c][f] = Wr[r][f][c]
__global__ __launch_bounds__(256) void basis_compose_kernel(const float* __restrict__ comp, const float* __restrict__ basis,
                                                            int R, int F, int O, float* __restrict__ Wr,
                                                            float* __restrict__ WrT) {
    __shared__ float s_comp[RR * NB];
    for (int i = threadIdx.x; i < R * NB; i += 256) s_comp[i] = comp[i];
    __syncthreads();
    const int FO = F * O, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= FO) return;
    float bv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bv[b] = basis[(int64_t)b * FO + k];
    const int f = k / O, c = k - f * O;
    for (int r = 0; r < R; ++r) {
        float a = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) a += s_comp[r * NB + b] * bv[b];
        Wr[(int64_t)r * FO + k] = a;
        WrT[(int64_t)r * FO + (int64_t)c * F + f] = a;
    }
}

// blocks [0, nA): dbasis[b][k] = sum_r comp[r,b] dWr[r][k];  blocks nA + (r*NB + b): dcomp[r,b] = <dWr[r], basis[b]>
__global__ __launch_bounds__(256) void basis_decompose_kernel(const float* __restrict__ comp, const float* __restrict__ basis,
                                                              const float* __restrict__ dWr, int R, int FO, int nA,
                                                              float* __restrict__ dbasis, float* __restrict__ dcomp) {
    __shared__ float s_comp[RR * NB];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nA) {
        for (int i = tid; i < R * NB; i += 256) s_comp[i] = comp[i];
        __syncthreads();
        const int k = blockIdx.x * 256 + tid;
        if (k >= FO) return;
        float g[RR];
#pragma unroll
        for (int r = 0; r < RR; ++r) g[r] = r < R ? dWr[(int64_t)r * FO + k] : 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < RR; ++r) a += (r < R ? s_comp[r * NB + b] : 0.f) * g[r];
            dbasis[(int64_t)b * FO + k] = a;
        }
        return;
    }
    const int rb = blockIdx.x - nA, r = rb / NB, b = rb - r * NB;
    const float* gw = dWr + (int64_t)r * FO;
    const float* bw = basis + (int64_t)b * FO;
    float a = 0.f;
    for (int k = tid; k < FO; k += 256) a += gw[k] * bw[k];
    a = wave_sum(a);
    if ((tid & 63) == 0) s_red[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) dcomp[rb] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// Z[i, r*F + c] = sum_{e into i, type_e == r} norm_e x[src_e, c]
__global__ __launch_bounds__(256) void rrgcn_agg_fwd_kernel(const float* __restrict__ x, int ldx, int F, int N, int R,
                                                            const int32_t* __restrict__ in_ptr,
                                                            const int32_t* __restrict__ in_src,
                                                            const int32_t* __restrict__ in_typ,
                                                            const float* __restrict__ norm, float* __restrict__ Z) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    float acc[RR][4];
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[r][u] = 0.f;
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    for (int w0 = e0; w0 < e1; w0 += 64) {      // lane-parallel edge metadata, then 8 source rows per batch
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_src = in_src[el], my_typ = in_typ[el];
        const float my_n = lane < nwin ? norm[el] : 0.f;
        for (int base = 0; base < nwin; base += EB) {
            Lane4 xs[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) xs[u] = load4(x + (int64_t)__shfl(my_src, min(base + u, nwin - 1), 64) * ldx, F, lane);
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int t = __shfl(my_typ, min(base + u, 63), 64);
                const float ne = __shfl(my_n, min(base + u, 63), 64);      // 0 past the window
#pragma unroll
                for (int r = 0; r < RR; ++r) {
                    const float c = t == r ? ne : 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[r][q] += c * xs[u].v[q];
                }
            }
        }
    }
    float* z = Z + (int64_t)i * R * F;
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r < R && lane + 64 * u < F) z[r * F + lane + 64 * u] = acc[r][u];
}

// dnorm_e = x[src_e] . dZ[i, type_e, :]  for every edge e into i; 16 edges per butterfly
__global__ __launch_bounds__(256) void rrgcn_bwd_target_kernel(const float* __restrict__ x, int ldx, int F, int N, int R,
                                                               const int32_t* __restrict__ in_ptr,
                                                               const int32_t* __restrict__ in_src,
                                                               const int32_t* __restrict__ in_typ,
                                                               const float* __restrict__ dZ, float* __restrict__ dnorm) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    Lane4 dz[RR];
    const float* z = dZ + (int64_t)i * R * F;
#pragma unroll
    for (int r = 0; r < RR; ++r) dz[r] = load4(z + (int64_t)min(r, R - 1) * F, F, lane);
    const int entry = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_src = in_src[el], my_typ = in_typ[el];
        for (int base = 0; base < nwin; base += 16) {
            float part[16];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Lane4 xs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    xs[u] = load4(x + (int64_t)__shfl(my_src, min(base + 8 * h + u, nwin - 1), 64) * ldx, F, lane);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = __shfl(my_typ, min(base + 8 * h + u, 63), 64);
                    float d = 0.f;
#pragma unroll
                    for (int r = 0; r < RR; ++r) d = t == r ? dot4(xs[u], dz[r]) : d;
                    part[8 * h + u] = d;
                }
            }
            const float tot = butterfly16_sum(part, lane);      // edge base + entry
            if ((lane & 3) == 0 && base + entry < nwin) dnorm[w0 + base + entry] = tot;
        }
    }
}

// U[j, r*O + c] = sum_{e out of j, type r} norm_e dH[dst_e, c]   (O <= 128)
__global__ __launch_bounds__(256) void rrgcn_bwd_source_kernel(const float* __restrict__ dH, int lddh, int O, int N, int R,
                                                               const int32_t* __restrict__ out_ptr,
                                                               const int32_t* __restrict__ out_dst,
                                                               const int32_t* __restrict__ out_typ,
                                                               const int32_t* __restrict__ out_eid,
                                                               const float* __restrict__ norm, float* __restrict__ U) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    float acc[RR][2];
#pragma unroll
    for (int r = 0; r < RR; ++r) acc[r][0] = acc[r][1] = 0.f;
    const int c0 = min(lane, O - 1), c1 = min(lane + 64, O - 1);
    const float m0 = lane < O ? 1.f : 0.f, m1 = lane + 64 < O ? 1.f : 0.f;
    const int e0 = out_ptr[j], e1 = out_ptr[j + 1];
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_dst = out_dst[el], my_typ = out_typ[el];
        const float my_n = lane < nwin ? norm[out_eid[el]] : 0.f;
        for (int base = 0; base < nwin; base += EB) {
            float g0[EB], g1[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const float* g = dH + (int64_t)__shfl(my_dst, min(base + u, nwin - 1), 64) * lddh;
                g0[u] = g[c0] * m0, g1[u] = g[c1] * m1;
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int t = __shfl(my_typ, min(base + u, 63), 64);
                const float ne = __shfl(my_n, min(base + u, 63), 64);
#pragma unroll
                for (int r = 0; r < RR; ++r) {
                    const float c = t == r ? ne : 0.f;
                    acc[r][0] += c * g0[u];
                    acc[r][1] += c * g1[u];
                }
            }
        }
    }
    float* u = U + (int64_t)j * R * O;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        if (r < R && lane < O) u[r * O + lane] = acc[r][0];
        if (r < R && lane + 64 < O) u[r * O + lane + 64] = acc[r][1];
    }
}

// out[n][c][r] = in[n][r][c]  (basis[b] [F,O] -> [O,F] so that dx = U [N,30*O] x basisT [30*O, F] is one GEMM)
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ in, int nb, int R, int Cc,
                                                                float* __restrict__ out) {
    const int64_t total = (int64_t)nb * R * Cc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cc);
        const int r = (int)((i / Cc) % R);
        const int64_t n = i / ((int64_t)Cc * R);
        out[(n * Cc + c) * R + r] = in[i];
    }
}

// out[i,:] (+)= sum_{e in row i of the CSR} x[idx[e],:]     (GraphConv aggregation and its transpose)
__global__ __launch_bounds__(256) void csr_sum_kernel(const float* __restrict__ x, int ldx, int F, int N,
                                                      const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                      float* __restrict__ out, int ldo, int accumulate) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    Lane4 acc = {{0.f, 0.f, 0.f, 0.f}};
    const int e0 = ptr[i], e1 = ptr[i + 1];
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int my_idx = idx[w0 + min(lane, nwin - 1)];
        for (int base = 0; base < nwin; base += EB) {
            Lane4 v[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) v[u] = load4(x + (int64_t)__shfl(my_idx, min(base + u, nwin - 1), 64) * ldx, F, lane);
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const float m = base + u < nwin ? 1.f : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc.v[q] += m * v[u].v[q];
            }
        }
    }
    float* o = out + (int64_t)i * ldo;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < F) o[lane + 64 * u] = accumulate ? o[lane + 64 * u] + acc.v[u] : acc.v[u];
}

}  // namespace

#define NODE_GRID(N) dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream

extern "C" int erc_gather_rows(const float* src, int lds, const int32_t* map, int N, int F, float* dst, int ldd,
                               int scatter, void* stream) {
    ERC_REQUIRE(src && map && dst && N > 0 && F > 0, "gather_rows: bad arguments");
    hipLaunchKernelGGL(gather_rows_kernel, NODE_GRID(N), src, lds, map, N, F, dst, ldd, scatter);
    ERC_LAUNCH_CHECK("gather_rows");
    return ERC_OK;
}

extern "C" int erc_edge_att_fwd(const float* x, int ldx, const float* att, int lda, int F, int N,
                                const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid, float* norm,
                                void* stream) {
    ERC_REQUIRE(x && att && out_ptr && out_dst && out_eid && norm, "edge_att_fwd: null pointer");
    ERC_REQUIRE(N > 0 && F > 0 && F <= 256, "edge_att_fwd: N=%d F=%d", N, F);
    hipLaunchKernelGGL(edge_att_fwd_kernel, NODE_GRID(N), x, ldx, att, lda, F, N, out_ptr, out_dst, out_eid, norm);
    ERC_LAUNCH_CHECK("edge_att_fwd");
    return ERC_OK;
}

extern "C" int erc_edge_att_bwd_parts(const float* x, int ldx, const float* att, int lda, int F, int N, const int32_t* in_ptr,
                                      const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                      const int32_t* out_eid, const float* norm, const float* dnorm, int dn_parts,
                                      int64_t dn_stride, float* dx, int lddx, int accumulate_dx, float* datt, int ldda,
                                      float* dscore, void* stream);

extern "C" int erc_edge_att_bwd_fused(const float* x, int ldx, const float* att, int lda, int F, int N, const int32_t* in_ptr,
                                      const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                      const int32_t* out_eid, const float* norm, const float* dnorm, int dn_parts,
                                      int64_t dn_stride, float* dx, int lddx, int accumulate_dx, float* datt, int ldda,
                                      float* dscore, const float* dx_slabs, int n_dx_slabs, int64_t dx_slab_stride,
                                      const float* rs_TT, const int32_t* rs_typ, const int32_t* rs_counts, float* rs_datt,
                                      int rs_R, void* stream);

extern "C" int erc_edge_att_bwd(const float* x, int ldx, const float* att, int lda, int F, int N, const int32_t* in_ptr,
                                const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                const int32_t* out_eid, const float* norm, const float* dnorm, float* dx, int lddx,
                                int accumulate_dx, float* datt, int ldda, float* dscore, void* stream) {
    return erc_edge_att_bwd_parts(x, ldx, att, lda, F, N, in_ptr, in_src, out_ptr, out_dst, out_eid, norm, dnorm, 1, 0, dx, lddx,
                                  accumulate_dx, datt, ldda, dscore, stream);
}

extern "C" int erc_edge_att_bwd_parts(const float* x, int ldx, const float* att, int lda, int F, int N, const int32_t* in_ptr,
                                      const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                      const int32_t* out_eid, const float* norm, const float* dnorm, int dn_parts,
                                      int64_t dn_stride, float* dx, int lddx, int accumulate_dx, float* datt, int ldda,
                                      float* dscore, void* stream) {
    ERC_REQUIRE(dn_parts >= 1 && (dn_parts == 1 || dn_stride > 0), "edge_att_bwd: dn_parts=%d", dn_parts);
    ERC_REQUIRE(x && att && in_ptr && in_src && out_ptr && out_dst && out_eid && norm && dnorm && dx && datt && dscore,
                "edge_att_bwd: null pointer");
    ERC_REQUIRE(N > 0 && F > 0 && F <= 256, "edge_att_bwd: N=%d F=%d", N, F);
    return erc_edge_att_bwd_fused(x, ldx, att, lda, F, N, in_ptr, in_src, out_ptr, out_dst, out_eid, norm, dnorm, dn_parts, dn_stride, dx,
                                  lddx, accumulate_dx, datt, ldda, dscore, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int erc_edge_att_bwd_fused(const float* x, int ldx, const float* att, int lda, int F, int N, const int32_t* in_ptr,
                                      const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                      const int32_t* out_eid, const float* norm, const float* dnorm, int dn_parts,
                                      int64_t dn_stride, float* dx, int lddx, int accumulate_dx, float* datt, int ldda,
                                      float* dscore, const float* dx_slabs, int n_dx_slabs, int64_t dx_slab_stride,
                                      const float* rs_TT, const int32_t* rs_typ, const int32_t* rs_counts, float* rs_datt,
                                      int rs_R, void* stream) {
    ERC_REQUIRE(dn_parts >= 1 && (dn_parts == 1 || dn_stride > 0), "edge_att_bwd: dn_parts=%d", dn_parts);
    ERC_REQUIRE(x && att && in_ptr && in_src && out_ptr && out_dst && out_eid && norm && dnorm && dx && datt && dscore,
                "edge_att_bwd: null pointer");
    ERC_REQUIRE(N > 0 && F > 0 && F <= 256, "edge_att_bwd: N=%d F=%d", N, F);
    ERC_REQUIRE(!dx_slabs || (n_dx_slabs >= 1 && dx_slab_stride >= (int64_t)N * F), "edge_att_bwd_fused: %d slabs of stride %lld",
                n_dx_slabs, (long long)dx_slab_stride);
    ERC_REQUIRE(!rs_TT || (rs_typ && rs_counts && rs_datt && rs_R > 0), "edge_att_bwd_fused: relation sums need typ, counts, datt, R");
    EdgeBwdExtra ex{};
    ex.dx_slabs = dx_slabs, ex.n_dx_slabs = n_dx_slabs, ex.dx_slab_stride = dx_slab_stride;
    ex.rs_TT = rs_TT, ex.rs_typ = rs_typ, ex.rs_counts = rs_counts, ex.rs_datt = rs_datt, ex.rs_R = rs_TT ? rs_R : 0;
    ex.node_blocks = (N + 3) / 4;
    hipLaunchKernelGGL(edge_att_bwd_source_kernel, dim3(ex.node_blocks + ex.rs_R), dim3(256), 0, (hipStream_t)stream, att, lda, F, N,
                       out_ptr, out_dst, out_eid, norm, dnorm, dx, lddx, dscore, accumulate_dx, dn_parts, dn_stride, ex);
    ERC_LAUNCH_CHECK("edge_att_bwd_source");
    hipLaunchKernelGGL(edge_att_bwd_target_kernel, NODE_GRID(N), x, ldx, F, N, in_ptr, in_src, dscore, datt, ldda);
    ERC_LAUNCH_CHECK("edge_att_bwd_target");
    return ERC_OK;
}

extern "C" int erc_brgcn_agg_fwd(const float* x, int ldx, int F, int N, const int32_t* in_ptr, const int32_t* in_src,
                                 const int32_t* in_typ, const float* norm, const float* att, int num_bases, float* Z,
                                 void* stream) {
    ERC_REQUIRE(x && in_ptr && in_src && in_typ && norm && att && Z, "brgcn_agg_fwd: null pointer");
    ERC_REQUIRE(num_bases == NB && N > 0 && F > 0 && F <= 256, "brgcn_agg_fwd: num_bases=%d (built for %d) F=%d", num_bases, NB, F);
    hipLaunchKernelGGL(brgcn_agg_fwd_kernel, NODE_GRID(N), x, ldx, F, N, in_ptr, in_src, in_typ, norm, att, Z);
    ERC_LAUNCH_CHECK("brgcn_agg_fwd");
    return ERC_OK;
}

extern "C" int erc_brgcn_bwd_edges(const float* x, int ldx, int F, int N, int R, const int32_t* in_ptr,
                                   const int32_t* in_src, const int32_t* in_typ, const int32_t* counts,
                                   const float* norm, const float* att, int num_bases, const float* dZ, float* dnorm,
                                   float* TT, float* datt, void* stream) {
    ERC_REQUIRE(x && in_ptr && in_src && in_typ && counts && norm && att && dZ && dnorm && TT && datt,
                "brgcn_bwd_edges: null pointer");
    ERC_REQUIRE(num_bases == NB && N > 0 && R > 0 && F > 0 && F <= 256, "brgcn_bwd_edges: bad sizes");
    hipLaunchKernelGGL(brgcn_bwd_target_kernel, NODE_GRID(N), x, ldx, F, N, in_ptr, in_src, in_typ, norm, att, dZ, dnorm,
                       TT);
    ERC_LAUNCH_CHECK("brgcn_bwd_target");
    hipLaunchKernelGGL(rel_sum_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, TT, in_typ, counts, datt);
    ERC_LAUNCH_CHECK("rel_sum");
    return ERC_OK;
}

extern "C" int erc_brgcn_bwd_source(const float* dH, int lddh, int O, int N, const int32_t* out_ptr,
                                    const int32_t* out_dst, const int32_t* out_typ, const int32_t* out_eid,
                                    const float* norm, const float* att, int num_bases, float* U, void* stream) {
    ERC_REQUIRE(dH && out_ptr && out_dst && out_typ && out_eid && norm && att && U, "brgcn_bwd_source: null pointer");
    ERC_REQUIRE(num_bases == NB && N > 0 && O > 0 && O <= 128, "brgcn_bwd_source: bad sizes");
    hipLaunchKernelGGL(brgcn_bwd_source_kernel, NODE_GRID(N), dH, lddh, O, N, out_ptr, out_dst, out_typ, out_eid, norm,
                       att, U);
    ERC_LAUNCH_CHECK("brgcn_bwd_source");
    return ERC_OK;
}

// diagnostic: 6 x uint64 phase stamps (10 ns ticks) of the middle tile's workgroup of basis group 2; nullptr = off
extern "C" int erc_brgcn_set_stamps(unsigned long long* stamps) {
    g_brgcn_stamps = stamps;
    return ERC_OK;
}

// the slab workspace of erc_brgcn_fwd_tile (floats) and its number of slabs
extern "C" int64_t erc_brgcn_fwd_tile_slab_floats(int n_nodes) { return (int64_t)NGRP * n_nodes * TO; }
extern "C" int erc_brgcn_fwd_tile_slabs(void) { return NGRP; }

extern "C" int erc_brgcn_fwd_tile(const float* x, int ldx, int F, int O, int N, const int32_t* in_ptr, const int32_t* in_src,
                                  const int32_t* in_typ, const float* norm, const float* att, int num_bases,
                                  const float* basis, const float* root, float* Z, float* slabs, void* stream) {
    ERC_REQUIRE(x && in_ptr && in_src && in_typ && norm && att && basis && root && Z && slabs, "brgcn_fwd_tile: null pointer");
    ERC_REQUIRE(num_bases == NB && F == TF && O == TO && N > 0 && ldx % 4 == 0,
                "brgcn_fwd_tile: built for %d bases, F = %d, O = %d (got %d, %d, %d)", NB, TF, TO, num_bases, F, O);
    ERC_REQUIRE(((uintptr_t)basis & 15) == 0 && ((uintptr_t)root & 15) == 0 && ((uintptr_t)slabs & 15) == 0,
                "brgcn_fwd_tile: basis / root / slabs must be 16-byte aligned");
    static bool attr_set = false;
    const int lds = 16 * TKP * (int)sizeof(float);
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)brgcn_fwd_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            erc_set_error("brgcn_fwd_tile: cannot reserve %d bytes of LDS", lds);
            return ERC_E_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(brgcn_fwd_tile_kernel, dim3(erc_cdiv(N, 16), NGRP), dim3(512), lds, (hipStream_t)stream, x, ldx, N, in_ptr,
                       in_src, in_typ, norm, att, basis, root, Z, slabs, g_brgcn_stamps);
    ERC_LAUNCH_CHECK("brgcn_fwd_tile");
    return ERC_OK;
}

extern "C" int erc_brgcn_bwd_source_tile(const float* dH, int lddh, int F, int O, int N, const int32_t* out_ptr,
                                         const int32_t* out_dst, const int32_t* out_typ, const int32_t* out_eid,
                                         const float* norm, const float* att, int num_bases, const float* basis,
                                         const float* root, float* slabs, void* stream) {
    ERC_REQUIRE(dH && out_ptr && out_dst && out_typ && out_eid && norm && att && basis && root && slabs,
                "brgcn_bwd_source_tile: null pointer");
    ERC_REQUIRE(num_bases == NB && F == TF && O == TO && N > 0, "brgcn_bwd_source_tile: built for %d bases, F = %d, O = %d",
                NB, TF, TO);
    ERC_REQUIRE(((uintptr_t)basis & 15) == 0 && ((uintptr_t)root & 15) == 0 && ((uintptr_t)slabs & 15) == 0,
                "brgcn_bwd_source_tile: basis / root / slabs must be 16-byte aligned");
    static bool attr_set = false;
    const int lds = 8 * 16 * 208 * (int)sizeof(float);       // the partial tiles (106 KB) outsize the U tile (39 KB)
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)brgcn_bwd_source_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            erc_set_error("brgcn_bwd_source_tile: cannot reserve %d bytes of LDS", lds);
            return ERC_E_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(brgcn_bwd_source_tile_kernel, dim3(erc_cdiv(N, 16), NGRP), dim3(512), lds, (hipStream_t)stream, dH, lddh,
                       N, out_ptr, out_dst, out_typ, out_eid, norm, att, basis, root, slabs);
    ERC_LAUNCH_CHECK("brgcn_bwd_source_tile");
    return ERC_OK;
}

extern "C" int erc_brgcn_bwd_edges_tile(const float* x, int ldx, int F, int O, int N, int R, const int32_t* in_ptr,
                                        const int32_t* in_src, const int32_t* in_typ, const int32_t* counts,
                                        const float* norm, const float* att, int num_bases, const float* basis,
                                        const float* dH, int lddh, float* TT, float* dn_slabs, int64_t dn_stride,
                                        float* datt, void* stream) {
    ERC_REQUIRE(x && in_ptr && in_src && in_typ && counts && norm && att && basis && dH && TT && dn_slabs,
                "brgcn_bwd_edges_tile: null pointer");
    ERC_REQUIRE(num_bases == NB && F == TF && O == TO && N > 0 && R > 0 && lddh % 4 == 0 && dn_stride > 0,
                "brgcn_bwd_edges_tile: built for %d bases, F = %d, O = %d", NB, TF, TO);
    ERC_REQUIRE(((uintptr_t)basis & 15) == 0 && ((uintptr_t)dH & 15) == 0, "brgcn_bwd_edges_tile: basis / dOut must be 16-byte aligned");
    static bool attr_set = false;
    const int lds = 16 * DZP * (int)sizeof(float);
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)brgcn_bwd_target_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            erc_set_error("brgcn_bwd_edges_tile: cannot reserve %d bytes of LDS", lds);
            return ERC_E_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(brgcn_bwd_target_tile_kernel, dim3(erc_cdiv(N, 16), NGRP), dim3(512), lds, (hipStream_t)stream, x, ldx, N,
                       in_ptr, in_src, in_typ, norm, att, basis, dH, lddh, TT, dn_slabs, dn_stride, g_brgcn_stamps ? g_brgcn_stamps + 8 : nullptr);
    ERC_LAUNCH_CHECK("brgcn_bwd_target_tile");
    if (datt) {      // (NULL: the caller sums TT per relation inside a later launch -- erc_edge_att_bwd_fused)
        hipLaunchKernelGGL(rel_sum_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, TT, in_typ, counts, datt);
        ERC_LAUNCH_CHECK("rel_sum");
    }
    return ERC_OK;
}

extern "C" int erc_rrgcn_max_relations(void) { return RR; }

extern "C" int erc_basis_compose(const float* comp, const float* basis, int R, int num_bases, int F, int O, float* Wr,
                                 float* WrT, void* stream) {
    ERC_REQUIRE(comp && basis && Wr && WrT, "basis_compose: null pointer");
    ERC_REQUIRE(num_bases == NB && R > 0 && R <= RR && F > 0 && O > 0, "basis_compose: R=%d (<= %d) num_bases=%d (built for %d)",
                R, RR, num_bases, NB);
    hipLaunchKernelGGL(basis_compose_kernel, dim3(erc_cdiv(F * O, 256)), dim3(256), 0, (hipStream_t)stream, comp, basis, R, F,
                       O, Wr, WrT);
    ERC_LAUNCH_CHECK("basis_compose");
    return ERC_OK;
}

extern "C" int erc_basis_decompose(const float* comp, const float* basis, const float* dWr, int R, int num_bases, int FO,
                                   float* dbasis, float* dcomp, void* stream) {
    ERC_REQUIRE(comp && basis && dWr && dbasis && dcomp, "basis_decompose: null pointer");
    ERC_REQUIRE(num_bases == NB && R > 0 && R <= RR && FO > 0, "basis_decompose: bad sizes");
    const int nA = erc_cdiv(FO, 256);
    hipLaunchKernelGGL(basis_decompose_kernel, dim3(nA + R * NB), dim3(256), 0, (hipStream_t)stream, comp, basis, dWr, R, FO,
                       nA, dbasis, dcomp);
    ERC_LAUNCH_CHECK("basis_decompose");
    return ERC_OK;
}

extern "C" int erc_rrgcn_agg_fwd(const float* x, int ldx, int F, int N, int R, const int32_t* in_ptr, const int32_t* in_src,
                                 const int32_t* in_typ, const float* norm, float* Z, void* stream) {
    ERC_REQUIRE(x && in_ptr && in_src && in_typ && norm && Z, "rrgcn_agg_fwd: null pointer");
    ERC_REQUIRE(R > 0 && R <= RR && N > 0 && F > 0 && F <= 256, "rrgcn_agg_fwd: R=%d (<= %d) F=%d", R, RR, F);
    hipLaunchKernelGGL(rrgcn_agg_fwd_kernel, NODE_GRID(N), x, ldx, F, N, R, in_ptr, in_src, in_typ, norm, Z);
    ERC_LAUNCH_CHECK("rrgcn_agg_fwd");
    return ERC_OK;
}

extern "C" int erc_rrgcn_bwd_edges(const float* x, int ldx, int F, int N, int R, const int32_t* in_ptr,
                                   const int32_t* in_src, const int32_t* in_typ, const float* dZ, float* dnorm,
                                   void* stream) {
    ERC_REQUIRE(x && in_ptr && in_src && in_typ && dZ && dnorm, "rrgcn_bwd_edges: null pointer");
    ERC_REQUIRE(R > 0 && R <= RR && N > 0 && F > 0 && F <= 256, "rrgcn_bwd_edges: bad sizes");
    hipLaunchKernelGGL(rrgcn_bwd_target_kernel, NODE_GRID(N), x, ldx, F, N, R, in_ptr, in_src, in_typ, dZ, dnorm);
    ERC_LAUNCH_CHECK("rrgcn_bwd_target");
    return ERC_OK;
}

extern "C" int erc_rrgcn_bwd_source(const float* dH, int lddh, int O, int N, int R, const int32_t* out_ptr,
                                    const int32_t* out_dst, const int32_t* out_typ, const int32_t* out_eid,
                                    const float* norm, float* U, void* stream) {
    ERC_REQUIRE(dH && out_ptr && out_dst && out_typ && out_eid && norm && U, "rrgcn_bwd_source: null pointer");
    ERC_REQUIRE(R > 0 && R <= RR && N > 0 && O > 0 && O <= 128, "rrgcn_bwd_source: bad sizes");
    hipLaunchKernelGGL(rrgcn_bwd_source_kernel, NODE_GRID(N), dH, lddh, O, N, R, out_ptr, out_dst, out_typ, out_eid, norm, U);
    ERC_LAUNCH_CHECK("rrgcn_bwd_source");
    return ERC_OK;
}

extern "C" int erc_transpose_batched(const float* in, int nb, int rows, int cols, float* out, void* stream) {
    ERC_REQUIRE(in && out && nb > 0 && rows > 0 && cols > 0, "transpose_batched: bad arguments");
    const int64_t total = (int64_t)nb * rows * cols;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, nb, rows, cols, out);
    ERC_LAUNCH_CHECK("transpose_batched");
    return ERC_OK;
}

extern "C" int erc_csr_sum(const float* x, int ldx, int F, int N, const int32_t* ptr, const int32_t* idx, float* out,
                           int ldo, int accumulate, void* stream) {
    ERC_REQUIRE(x && ptr && idx && out && N > 0 && F > 0 && F <= 256, "csr_sum: bad arguments");
    hipLaunchKernelGGL(csr_sum_kernel, NODE_GRID(N), x, ldx, F, N, ptr, idx, out, ldo, accumulate);
    ERC_LAUNCH_CHECK("csr_sum");
    return ERC_OK;
}
