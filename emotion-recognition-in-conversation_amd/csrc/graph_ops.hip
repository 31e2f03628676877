// K3 / K4: message passing over the dialogue graph, one wavefront per node.
//
// Both operators are written as GATHERS over a CSR (by target for the forward
// and the target-side backward, by source for the source-side backward), so
// every output row is produced by exactly one wavefront in a fixed edge order:
// deterministic, no float atomics.  Feature rows are F <= 128 floats; lane l
// owns channels l and l+64.
#include "erc_common.h"

namespace {

constexpr int MAX_R = 8;
constexpr int CH = 12; // neighbour rows fetched together (all loads in flight before the first use); the COGMEN window
                       // graph (wp = wf = 5) has at most 11 in-edges per node: one batch

// Every kernel below follows the same latency discipline (one dependent global round trip costs ~0.5 us here,
// a 20-edge neighbourhood walked edge by edge ~20 of them): the CSR slice of the node is loaded lane-parallel
// (lane l holds edge e0+l of a 64-edge window), indices are broadcast with __shfl, and the neighbour rows of CH
// edges are fetched by unconditional loads (clamped indices, results masked afterwards) before any arithmetic.
struct Row2 {
    float a, b;
};
__device__ __forceinline__ Row2 ldrow(const float* base, int64_t row, int ld, int c0, int c1) {
    const float* p = base + row * ld;
    Row2 r;
    r.a = p[c0];
    r.b = p[c1];
    return r;
}

// ---------------------------------------------------------------- RGCN mean
__global__ __launch_bounds__(256) void rgcn_mean_fwd_kernel(const float* __restrict__ x, int ldx, int F, int R, int N,
                                                            const int32_t* __restrict__ in_ptr,
                                                            const int32_t* __restrict__ in_src,
                                                            const int32_t* __restrict__ in_typ,
                                                            float* __restrict__ Mout, int ldm,
                                                            float* __restrict__ inv_cnt) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const bool h0 = lane < F, h1 = lane + 64 < F;
    const int c0 = min(lane, F - 1), c1 = min(lane + 64, F - 1);
    float s0[MAX_R], s1[MAX_R];
    int cnt[MAX_R];
#pragma unroll
    for (int r = 0; r < MAX_R; ++r) s0[r] = s1[r] = 0.f, cnt[r] = 0;
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    const Row2 self = ldrow(x, i, ldx, c0, c1);
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_src = in_src[el], my_typ = in_typ[el];
        for (int base = 0; base < nwin; base += CH) {
            Row2 v[CH];
            int t[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int sl = min(base + u, nwin - 1);
                v[u] = ldrow(x, __shfl(my_src, sl, 64), ldx, c0, c1);
                t[u] = (base + u < nwin) ? __shfl(my_typ, sl, 64) : -1;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int r = 0; r < MAX_R; ++r) {
                    const bool hit = (t[u] == r);
                    s0[r] += hit ? v[u].a : 0.f;
                    s1[r] += hit ? v[u].b : 0.f;
                    cnt[r] += hit ? 1 : 0;
                }
        }
    }
    float* mrow = Mout + (int64_t)i * ldm;
#pragma unroll
    for (int r = 0; r < MAX_R; ++r) {
        if (r < R) {
            // mean = sum / count (a division, like scatter-mean); 1/count is kept for the backward
            if (h0) mrow[r * F + lane] = cnt[r] > 0 ? s0[r] / (float)cnt[r] : 0.f;
            if (h1) mrow[r * F + lane + 64] = cnt[r] > 0 ? s1[r] / (float)cnt[r] : 0.f;
            if (lane == 0) inv_cnt[(int64_t)i * R + r] = cnt[r] > 0 ? 1.0f / (float)cnt[r] : 0.f;
        }
    }
    if (h0) mrow[R * F + lane] = self.a;
    if (h1) mrow[R * F + lane + 64] = self.b;
}

__global__ __launch_bounds__(256) void rgcn_mean_bwd_kernel(const float* __restrict__ dM, int ldm, int F, int R, int N,
                                                            const int32_t* __restrict__ out_ptr,
                                                            const int32_t* __restrict__ out_dst,
                                                            const int32_t* __restrict__ out_typ,
                                                            const float* __restrict__ inv_cnt,
                                                            float* __restrict__ dx, int lddx) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    const bool h0 = lane < F, h1 = lane + 64 < F;
    const int c0 = min(lane, F - 1), c1 = min(lane + 64, F - 1);
    const Row2 self = ldrow(dM + R * F, j, ldm, c0, c1);
    float a0 = self.a, a1 = self.b;
    const int e0 = out_ptr[j], e1 = out_ptr[j + 1];
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_dst = out_dst[el];
        const int my_typ = min(out_typ[el], R);           // relation ids >= R are ignored (weight 0 below)
        const float my_w = my_typ < R ? inv_cnt[(int64_t)my_dst * R + my_typ] : 0.f;
        const int my_off = min(my_typ, R - 1) * F;
        for (int base = 0; base < nwin; base += CH) {
            Row2 v[CH];
            float w[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int sl = min(base + u, nwin - 1);
                v[u] = ldrow(dM + __shfl(my_off, sl, 64), __shfl(my_dst, sl, 64), ldm, c0, c1);
                w[u] = (base + u < nwin) ? __shfl(my_w, sl, 64) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) a0 += v[u].a * w[u], a1 += v[u].b * w[u];
        }
    }
    if (h0) dx[(int64_t)j * lddx + lane] = a0;
    if (h1) dx[(int64_t)j * lddx + lane + 64] = a1;
}

// ------------------------------------------------- TransformerConv, heads=1
// CH (<= 16) per-lane partial dot products -> every lane gets all CH wave totals: one 16-value butterfly (17 shuffles)
// and CH lane reads instead of CH full wave reductions (6 shuffles each).
__device__ __forceinline__ void wave_sums_ch(const float (&part)[16], float (&tot)[CH], int lane) {
    static_assert(CH <= 16, "one butterfly");
    float b[8], c[4], d[2];
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (h5 ? part[8 + j] : part[j]) + __shfl_xor(h5 ? part[j] : part[8 + j], 32, 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = (h4 ? b[4 + j] : b[j]) + __shfl_xor(h4 ? b[j] : b[4 + j], 16, 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) d[j] = (h3 ? c[2 + j] : c[j]) + __shfl_xor(h3 ? c[j] : c[2 + j], 8, 64);
    float e = (h2 ? d[1] : d[0]) + __shfl_xor(h2 ? d[0] : d[1], 4, 64);
    e += __shfl_xor(e, 2, 64);
    e += __shfl_xor(e, 1, 64);      // lane l: total of entry ((l>>5)&1)*8 + ((l>>4)&1)*4 + ((l>>3)&1)*2 + ((l>>2)&1)
#pragma unroll
    for (int u = 0; u < CH; ++u) tot[u] = __shfl(e, ((u >> 3) & 1) * 32 + ((u >> 2) & 1) * 16 + ((u >> 1) & 1) * 8 + (u & 1) * 4, 64);
}

__global__ __launch_bounds__(256) void tconv_fwd_kernel(const float* __restrict__ qkvs, int ld, int F, int N,
                                                        float scale, const int32_t* __restrict__ in_ptr,
                                                        const int32_t* __restrict__ in_src, float* __restrict__ out,
                                                        int ldo, float* __restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const bool h0 = lane < F, h1 = lane + 64 < F;
    const int c0 = min(lane, F - 1), c1 = min(lane + 64, F - 1);
    const float m0 = h0 ? 1.f : 0.f, m1 = h1 ? 1.f : 0.f;
    const Row2 q = ldrow(qkvs, i, ld, c0, c1);
    const Row2 sk = ldrow(qkvs + 3 * F, i, ld, c0, c1);
    const float q0 = q.a * m0, q1 = q.b * m1;
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    // online softmax over 64-edge windows (one window for window graphs): running max mx, denominator den,
    // unnormalised output o; lane l keeps exp-score of edge l of the current window until it is rescaled.
    float mx = -INFINITY, den = 0.f, o0 = 0.f, o1 = 0.f;
    if (e1 - e0 <= CH && e1 > e0) {   // whole neighbourhood in one batch (wave-uniform): key AND value rows requested together
        const int nwin = e1 - e0;
        const int my_src = in_src[e0 + min(lane, max(nwin - 1, 0))];
        Row2 k[CH], v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int sj = __shfl(my_src, min(u, max(nwin - 1, 0)), 64);
            k[u] = ldrow(qkvs + F, sj, ld, c0, c1);
            v[u] = ldrow(qkvs + 2 * F, sj, ld, c0, c1);
        }
        float sc[CH], part[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) part[u] = u < CH ? q0 * k[u < CH ? u : 0].a + q1 * k[u < CH ? u : 0].b : 0.f;
        wave_sums_ch(part, sc, lane);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            sc[u] *= scale;
            if (u < nwin) mx = fmaxf(mx, sc[u]);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const float pw = u < nwin ? expf(sc[u] - mx) : 0.f;
            sc[u] = pw;
            den += pw;
            o0 += pw * v[u].a, o1 += pw * v[u].b;
        }
        const float inv = 1.0f / (den + 1e-16f);
        if (h0) out[(int64_t)i * ldo + lane] = o0 * inv + sk.a;
        if (h1) out[(int64_t)i * ldo + lane + 64] = o1 * inv + sk.b;
        float mine = 0.f;
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (lane == u) mine = sc[u];
        if (lane < nwin) alpha[e0 + lane] = mine * inv;
        return;
    }
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int my_src = in_src[w0 + min(lane, nwin - 1)];
        float my_s = -INFINITY;  // raw score of edge `lane` of this window
        for (int base = 0; base < nwin; base += CH) {
            Row2 k[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) k[u] = ldrow(qkvs + F, __shfl(my_src, min(base + u, nwin - 1), 64), ld, c0, c1);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float sc = wave_sum(q0 * k[u].a + q1 * k[u].b) * scale;
                if (lane == base + u) my_s = sc;
            }
        }
        const float wmx = wave_max(my_s);           // lanes >= nwin hold -inf
        const float nmx = fmaxf(mx, wmx);
        const float resc = expf(mx - nmx);          // 0 on the first window (mx = -inf)
        den *= resc, o0 *= resc, o1 *= resc;
        mx = nmx;
        const float my_p = lane < nwin ? expf(my_s - mx) : 0.f;
        den += wave_sum(my_p);
        for (int base = 0; base < nwin; base += CH) {
            Row2 v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = ldrow(qkvs + 2 * F, __shfl(my_src, min(base + u, nwin - 1), 64), ld, c0, c1);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float pw = (base + u < nwin) ? __shfl(my_p, min(base + u, 63), 64) : 0.f;
                o0 += pw * v[u].a, o1 += pw * v[u].b;
            }
        }
        // provisional weights (exact when there is a single window; rescaled below otherwise)
        if (lane < nwin) alpha[w0 + lane] = my_p;
        if (w0 > e0) {  // earlier windows were written relative to an older max
            // (degree > 64 only) bring them to the current max
            for (int e = e0 + lane; e < w0; e += 64) alpha[e] *= resc;
        }
    }
    const float inv = 1.0f / (den + 1e-16f);
    if (h0) out[(int64_t)i * ldo + lane] = o0 * inv + sk.a;
    if (h1) out[(int64_t)i * ldo + lane + 64] = o1 * inv + sk.b;
    // normalise: each lane rescales the entries it wrote itself (same lane -> same addresses, program order holds)
    for (int e = e0 + lane; e < e1; e += 64) alpha[e] *= inv;
}

// BN = true: the gradient wrt this layer's output is not given but derived on the fly from the BatchNorm that follows
// (elementwise part of BatchNorm1d's backward, cogmen.py:67,72): dout = gamma * rstd * (dY - ma - xhat * mb), with
// `dout` = dY, x = the BatchNorm input (= this layer's forward output), and written to dout_store for the source pass.
struct BnBwd {
    const float* x;       // [N, ldx]
    const float* gamma;   // [F]
    const float* saved;   // [0,F) mean, [F,2F) rstd
    const float* bn_bwd;  // [0,F) mean of dY, [F,2F) mean of dY * xhat
    float* dout_store;    // [N, lddo]
    int ldx;
};

template <bool BN>
__global__ __launch_bounds__(256) void tconv_bwd_target_kernel(const float* __restrict__ qkvs, int ld, int F, int N,
                                                               float scale, const int32_t* __restrict__ in_ptr,
                                                               const int32_t* __restrict__ in_src,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ dout, int lddo,
                                                               float* __restrict__ dqkvs,
                                                               float* __restrict__ dscore, const BnBwd bn) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const bool h0 = lane < F, h1 = lane + 64 < F;
    const int c0 = min(lane, F - 1), c1 = min(lane + 64, F - 1);
    Row2 gr = ldrow(dout, i, lddo, c0, c1);
    if (BN) {
        const Row2 xr = ldrow(bn.x, i, bn.ldx, c0, c1);
        const float mu0 = bn.saved[c0], mu1 = bn.saved[c1], rs0 = bn.saved[F + c0], rs1 = bn.saved[F + c1];
        const float ga0 = bn.gamma[c0], ga1 = bn.gamma[c1];
        const float ma0 = bn.bn_bwd[c0], ma1 = bn.bn_bwd[c1], mb0 = bn.bn_bwd[F + c0], mb1 = bn.bn_bwd[F + c1];
        gr.a = ga0 * rs0 * (gr.a - ma0 - (xr.a - mu0) * rs0 * mb0);
        gr.b = ga1 * rs1 * (gr.b - ma1 - (xr.b - mu1) * rs1 * mb1);
        if (h0) bn.dout_store[(int64_t)i * lddo + lane] = gr.a;
        if (h1) bn.dout_store[(int64_t)i * lddo + lane + 64] = gr.b;
    }
    const float g0 = h0 ? gr.a : 0.f, g1 = h1 ? gr.b : 0.f;
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    if (e1 - e0 <= CH && e1 > e0) {   // whole neighbourhood in one batch: value and key rows requested together, one pass
        const int nwin = e1 - e0;
        const int el = e0 + min(lane, nwin - 1);
        const int my_src = in_src[el];
        const float my_al = lane < nwin ? alpha[el] : 0.f;
        Row2 v[CH], kk[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int sj = __shfl(my_src, min(u, nwin - 1), 64);
            v[u] = ldrow(qkvs + 2 * F, sj, ld, c0, c1);
            kk[u] = ldrow(qkvs + F, sj, ld, c0, c1);
        }
        float da[CH], al[CH], part[16], t = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) part[u] = u < CH ? g0 * v[u < CH ? u : 0].a + g1 * v[u < CH ? u : 0].b : 0.f;
        wave_sums_ch(part, da, lane);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            al[u] = u < nwin ? __shfl(my_al, u, 64) : 0.f;
            t += al[u] * da[u];
        }
        float dq0 = 0.f, dq1 = 0.f, mine = 0.f;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const float ds = al[u] * (da[u] - t) * scale;
            dq0 += ds * kk[u].a, dq1 += ds * kk[u].b;
            if (lane == u) mine = ds;
        }
        if (lane < nwin) dscore[el] = mine;
        float* d = dqkvs + (int64_t)i * ld;
        if (h0) d[lane] = dq0, d[3 * F + lane] = g0;
        if (h1) d[lane + 64] = dq1, d[3 * F + lane + 64] = g1;
        return;
    }
    // pass 1: t = sum_e alpha_e (dout_i . v_src(e)); lane l keeps d alpha of edge l (first window)
    float t = 0.f;
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_src = in_src[el];
        const float my_al = lane < nwin ? alpha[el] : 0.f;
        float my_da = 0.f;
        for (int base = 0; base < nwin; base += CH) {
            Row2 v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = ldrow(qkvs + 2 * F, __shfl(my_src, min(base + u, nwin - 1), 64), ld, c0, c1);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float da = wave_sum(g0 * v[u].a + g1 * v[u].b);
                if (lane == base + u) my_da = da;
            }
        }
        t += wave_sum(my_al * my_da);
    }
    // pass 2: ds_e = alpha_e (d alpha_e - t) scale ; dq_i = sum_e ds_e k_src(e)
    float dq0 = 0.f, dq1 = 0.f;
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_src = in_src[el];
        const float my_al = lane < nwin ? alpha[el] : 0.f;
        float my_da = 0.f;
        Row2 kk[CH];
        for (int base = 0; base < nwin; base += CH) {
            Row2 v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = ldrow(qkvs + 2 * F, __shfl(my_src, min(base + u, nwin - 1), 64), ld, c0, c1);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float da = wave_sum(g0 * v[u].a + g1 * v[u].b);
                if (lane == base + u) my_da = da;
            }
        }
        const float my_ds = my_al * (my_da - t) * scale;
        if (lane < nwin) dscore[el] = my_ds;
        for (int base = 0; base < nwin; base += CH) {
#pragma unroll
            for (int u = 0; u < CH; ++u) kk[u] = ldrow(qkvs + F, __shfl(my_src, min(base + u, nwin - 1), 64), ld, c0, c1);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float ds = (base + u < nwin) ? __shfl(my_ds, min(base + u, 63), 64) : 0.f;
                dq0 += ds * kk[u].a, dq1 += ds * kk[u].b;
            }
        }
    }
    float* d = dqkvs + (int64_t)i * ld;
    if (h0) d[lane] = dq0, d[3 * F + lane] = g0;
    if (h1) d[lane + 64] = dq1, d[3 * F + lane + 64] = g1;
}

__global__ __launch_bounds__(256) void tconv_bwd_source_kernel(const float* __restrict__ qkvs, int ld, int F, int N,
                                                               const int32_t* __restrict__ out_ptr,
                                                               const int32_t* __restrict__ out_dst,
                                                               const int32_t* __restrict__ out_eid,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ dscore,
                                                               const float* __restrict__ dout, int lddo,
                                                               float* __restrict__ dqkvs) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    const bool h0 = lane < F, h1 = lane + 64 < F;
    const int c0 = min(lane, F - 1), c1 = min(lane + 64, F - 1);
    float dk0 = 0.f, dk1 = 0.f, dv0 = 0.f, dv1 = 0.f;
    const int e0 = out_ptr[j], e1 = out_ptr[j + 1];
    for (int w0 = e0; w0 < e1; w0 += 64) {
        const int nwin = min(64, e1 - w0);
        const int el = w0 + min(lane, nwin - 1);
        const int my_dst = out_dst[el];
        const int id = out_eid[el];
        const float my_ds = lane < nwin ? dscore[id] : 0.f;
        const float my_al = lane < nwin ? alpha[id] : 0.f;
        for (int base = 0; base < nwin; base += CH) {
            Row2 qv[CH], gv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int dst = __shfl(my_dst, min(base + u, nwin - 1), 64);
                qv[u] = ldrow(qkvs, dst, ld, c0, c1);
                gv[u] = ldrow(dout, dst, lddo, c0, c1);
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int sl = min(base + u, 63);
                const float ds = (base + u < nwin) ? __shfl(my_ds, sl, 64) : 0.f;
                const float al = (base + u < nwin) ? __shfl(my_al, sl, 64) : 0.f;
                dk0 += ds * qv[u].a, dk1 += ds * qv[u].b;
                dv0 += al * gv[u].a, dv1 += al * gv[u].b;
            }
        }
    }
    float* d = dqkvs + (int64_t)j * ld;
    if (h0) d[F + lane] = dk0, d[2 * F + lane] = dv0;
    if (h1) d[F + lane + 64] = dk1, d[2 * F + lane + 64] = dv1;
}

}  // namespace

extern "C" int erc_rgcn_mean_fwd(const float* x, int ldx, int F, int R, int N, const int32_t* in_ptr,
                                 const int32_t* in_src, const int32_t* in_typ, float* Mout, int ldm, float* inv_cnt,
                                 void* stream) {
    ERC_REQUIRE(x && in_ptr && in_src && in_typ && Mout && inv_cnt, "rgcn_mean_fwd: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && R > 0 && R <= MAX_R && N > 0, "rgcn_mean_fwd: F=%d R=%d N=%d unsupported", F, R, N);
    ERC_REQUIRE(ldm >= (R + 1) * F && ldx >= F, "rgcn_mean_fwd: leading dimension too small");
    hipLaunchKernelGGL(rgcn_mean_fwd_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, F, R, N,
                       in_ptr, in_src, in_typ, Mout, ldm, inv_cnt);
    ERC_LAUNCH_CHECK("rgcn_mean_fwd");
    return ERC_OK;
}

extern "C" int erc_rgcn_mean_bwd(const float* dM, int ldm, int F, int R, int N, const int32_t* out_ptr,
                                 const int32_t* out_dst, const int32_t* out_typ, const float* inv_cnt, float* dx,
                                 int lddx, void* stream) {
    ERC_REQUIRE(dM && out_ptr && out_dst && out_typ && inv_cnt && dx, "rgcn_mean_bwd: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && R > 0 && R <= MAX_R && N > 0, "rgcn_mean_bwd: F=%d R=%d N=%d unsupported", F, R, N);
    hipLaunchKernelGGL(rgcn_mean_bwd_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, dM, ldm, F, R, N,
                       out_ptr, out_dst, out_typ, inv_cnt, dx, lddx);
    ERC_LAUNCH_CHECK("rgcn_mean_bwd");
    return ERC_OK;
}

extern "C" int erc_tconv_attn_fwd(const float* qkvs, int ld, int F, int N, float scale, const int32_t* in_ptr,
                                  const int32_t* in_src, float* out, int ldo, float* alpha, void* stream) {
    ERC_REQUIRE(qkvs && in_ptr && in_src && out && alpha, "tconv_attn_fwd: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && N > 0 && ld >= 4 * F, "tconv_attn_fwd: F=%d N=%d ld=%d unsupported", F, N, ld);
    hipLaunchKernelGGL(tconv_fwd_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, qkvs, ld, F, N, scale,
                       in_ptr, in_src, out, ldo, alpha);
    ERC_LAUNCH_CHECK("tconv_attn_fwd");
    return ERC_OK;
}

extern "C" int erc_tconv_attn_bwd_target(const float* qkvs, int ld, int F, int N, float scale, const int32_t* in_ptr,
                                         const int32_t* in_src, const float* alpha, const float* dout, int lddo,
                                         float* dqkvs, float* dscore, const float* bn_x, int bn_ldx, const float* bn_gamma,
                                         const float* bn_saved, const float* bn_bwd, float* dout_store, void* stream) {
    ERC_REQUIRE(qkvs && in_ptr && in_src && alpha && dout && dqkvs && dscore, "tconv_attn_bwd_target: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && N > 0 && ld >= 4 * F, "tconv_attn_bwd_target: bad sizes");
    const BnBwd bn{bn_x, bn_gamma, bn_saved, bn_bwd, dout_store, bn_ldx};
    if (bn_x) {
        ERC_REQUIRE(bn_gamma && bn_saved && bn_bwd && dout_store && bn_ldx >= F, "tconv_attn_bwd_target: BatchNorm prologue operands");
        hipLaunchKernelGGL(tconv_bwd_target_kernel<true>, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, qkvs, ld, F,
                           N, scale, in_ptr, in_src, alpha, dout, lddo, dqkvs, dscore, bn);
    } else {
        hipLaunchKernelGGL(tconv_bwd_target_kernel<false>, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, qkvs, ld,
                           F, N, scale, in_ptr, in_src, alpha, dout, lddo, dqkvs, dscore, bn);
    }
    ERC_LAUNCH_CHECK("tconv_attn_bwd_target");
    return ERC_OK;
}

extern "C" int erc_tconv_attn_bwd_source(const float* qkvs, int ld, int F, int N, const int32_t* out_ptr,
                                         const int32_t* out_dst, const int32_t* out_eid, const float* alpha,
                                         const float* dscore, const float* dout, int lddo, float* dqkvs,
                                         void* stream) {
    ERC_REQUIRE(qkvs && out_ptr && out_dst && out_eid && alpha && dscore && dout && dqkvs,
                "tconv_attn_bwd_source: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && N > 0 && ld >= 4 * F, "tconv_attn_bwd_source: bad sizes");
    hipLaunchKernelGGL(tconv_bwd_source_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, qkvs, ld, F, N,
                       out_ptr, out_dst, out_eid, alpha, dscore, dout, lddo, dqkvs);
    ERC_LAUNCH_CHECK("tconv_attn_bwd_source");
    return ERC_OK;
}
