// COGMEN bf16 compute mode, first launch of the step: the input projection H0 = X[node rows] W1^T + b1
// (track_mm/cogmen.py:103-105,147: rnn.1 = nn.Linear(D, 100) on the valid utterances) AND the window graph of
// track_mm/cogmen_utils.py:109-172 -- every output of erc_window_graph_build (csrc/graph_build.hip) -- straight from the
// dialogue lengths and the speaker tensor.  The separate graph-build launch (5 us + a launch gap of a 100 us step) is gone:
//
//  * every workgroup scans the B lengths itself (one load per thread + a block scan: node and edge offsets of every
//    dialogue in LDS) -- the projection's feature rows are then addressed as (dialogue, position) with NO load of a
//    node -> row map in front of them: lengths -> rows is the same dependent depth as node_row -> rows was;
//  * the projection is the persistent kernel of csrc/gemm_stream.hip (weights resident in registers, 16-row groups, two
//    workgroups per row group on one XCD taking column tiles 0-3 / 4-6);
//  * the workgroup of a pair that owns column tiles 0-3 also writes the graph of its 16 nodes: one thread per (node,
//    in-edge slot | out-edge slot), all offsets from closed forms (erc_window_prefix), the neighbours' speakers requested
//    together with the feature rows.
// Bit-identical to erc_window_graph_build + erc_gemm_bf16a_stream with the gather (tests/test_gpu_cogmen_fused.py).
//
// SPLIT COMPUTE MODES (NT = 2, 3; csrc/split_dev.h): the feature block stays fp32 in memory (what the reference feeds
// rnn.1), every A fragment is expanded into NT bf16 terms in registers and multiplied with the NT term planes of the weight
// shadow -- fp32-class H0 on the bf16 matrix cores.  The resident weight fragments are NT times as many registers, so a row
// group is shared by 4 (NT = 2: two column tiles each) or 8 (NT = 3: one each) workgroups of one XCD instead of 2.
#include "erc_common.h"
#include "split_dev.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int PG_NB = 6;       // K blocks of 32 per wavefront (8 wavefronts: K <= 1536)
constexpr int PG_BMAX = 2048;  // dialogues per batch (offsets in LDS)

struct PgP {
    const void* X;                 // bf16 (NT = 1) / fp32 (split modes) [B*T, ldx]: the padded feature block
    const unsigned short* W;       // bf16 [NO, ldw]: shadow of rnn.1.weight; split modes: NT term planes, w_plane elements apart
    int64_t w_plane;
    const float* bias;             // [NO]
    float* H0;                     // out [N, ldh0]
    int ldx, ldw, ldh0, K, NO;
    const int64_t* lengths;        // [B]
    const int64_t* speakers;       // [B, T] strided
    int64_t spk_sb, spk_st;
    int B, T, wp, wf, S, n_cap, e_cap;
    int32_t *node_off, *node_row, *node_spk, *in_ptr, *in_src, *in_typ, *out_ptr, *out_dst, *out_typ, *out_eid, *counts;
    // RESIDENT mode (or null): the batch is a list of dialogues of a feature store that lives in HBM -- desc[b] = length of
    // dialogue slot b (0: empty slot), desc[B + b] = its first row in the store.  X / speakers are then the store's [U, ldx]
    // feature rows / [U] speaker ids, lengths is unused, and node_row holds store rows: no padded [B, T, D] block exists.
    const int32_t* desc;
};

// NT: bf16 terms per operand value (1: the bf16 compute mode); PG_NT: column tiles of 16 per workgroup; NWG: workgroups that
// share a row group (ids id, id + 8, ..: one XCD), NWG * PG_NT >= 7
template <int NT, int PG_NT, int NWG>
__global__ __launch_bounds__(512) void cogmen_project_graph_kernel(const PgP p) {
    __shared__ float red[8 * PG_NT * 4 * 64];   // partial tiles of the 8 wavefronts (32 KB at 4 column tiles)
    __shared__ int s_noff[PG_BMAX + 1], s_eoff[PG_BMAX + 1], s_base[PG_BMAX];   // node / edge offsets, first feature row of a dialogue
    __shared__ int s_wn[8], s_we[8];
    __shared__ int s_dlg[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int id = blockIdx.x, half = (id >> 3) % NWG, pair = (id & 7) + 8 * (id / (8 * NWG)), n_pairs = (int)gridDim.x / NWG;
    const int n_base = half * 16 * PG_NT;            // first column of this workgroup
    const int nkb = (p.K + 31) / 32;

    // ---- the lengths first: everything but the weight fragments waits for them
    int L0 = p.desc ? p.desc[min(tid, p.B - 1)] : (int)p.lengths[min(tid, p.B - 1)];
    int base0 = p.desc ? p.desc[p.B + min(tid, p.B - 1)] : min(tid, p.B - 1) * p.T;
    __builtin_amdgcn_sched_barrier(0);
    // ---- this wavefront's K blocks of W (as gemm_bf16a_persist_kernel): a block that would run past K is shifted back to
    //      end at K; the k it then shares with the previous block are zeroed in the W fragment (K >= 32, K % 4 == 0)
    int k0[PG_NB];
    u32x4 wraw[PG_NB][PG_NT][NT];
#pragma unroll
    for (int s = 0; s < PG_NB; ++s) {
        const int kb = w + 8 * s;
        k0[s] = kb < nkb ? min(kb * 32, p.K - 32) : 0;
#pragma unroll
        for (int nt = 0; nt < PG_NT; ++nt) {
            const int n = n_base + 16 * nt + r;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const unsigned short* wr = p.W + t * p.w_plane + (int64_t)min(n, p.NO - 1) * p.ldw + k0[s] + 8 * g;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(wr), hi = *reinterpret_cast<const u32x2*>(wr + 4);
                wraw[s][nt][t] = (u32x4){lo.x, lo.y, hi.x, hi.y};
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- exclusive scan of (nodes, edges) per dialogue over the batch, 512 dialogues per pass.  The thread of dialogue t
    //      knows its node range without reading LDS, so it marks the rows of this workgroup's FIRST row group right away:
    //      two barriers between the lengths and the feature-row addresses
    int carry_n = 0, carry_e = 0;
    const int n0_first = pair * 16;
    for (int c0 = 0; c0 < p.B; c0 += 512) {
        const int t = c0 + tid;
        int L = c0 == 0 ? L0 : (p.desc ? p.desc[min(t, p.B - 1)] : (int)p.lengths[min(t, p.B - 1)]);
        const int base = c0 == 0 ? base0 : (p.desc ? p.desc[p.B + min(t, p.B - 1)] : min(t, p.B - 1) * p.T);
        L = t < p.B ? min(max(L, 0), p.T) : 0;
        const int E = erc_window_prefix(L, L, p.wf, p.wp);
        int sn = L, se = E;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int vn = __shfl_up(sn, o, 64), ve = __shfl_up(se, o, 64);
            if (lane >= o) sn += vn, se += ve;
        }
        if (c0 > 0) __syncthreads();    // the previous pass has read s_wn / s_we
        if (lane == 63) s_wn[w] = sn, s_we[w] = se;
        __syncthreads();
        int bn = carry_n, be = carry_e, tn = 0, te = 0;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
            const int a = s_wn[ww], b = s_we[ww];
            tn += a, te += b;
            if (ww < w) bn += a, be += b;
        }
        if (t < p.B) {
            const int off = bn + sn - L;
            s_noff[t] = off, s_eoff[t] = be + se - E, s_base[t] = base;
            for (int n = max(off, n0_first); n < min(off + L, n0_first + 16); ++n) s_dlg[n - n0_first] = t;
        }
        carry_n += tn, carry_e += te;
    }
    if (tid == 0) s_noff[p.B] = carry_n, s_eoff[p.B] = carry_e;
    const int N = carry_n, Etot = carry_e;
    __syncthreads();
    const bool cap_ok = N <= p.n_cap && Etot <= p.e_cap;     // capacity guard (uniform); the host checks counts
    if (id == 0) {
        for (int t = tid; t <= p.B; t += 512) p.node_off[t] = s_noff[t];
        if (tid == 0) {
            p.counts[0] = N, p.counts[1] = Etot;
            if (cap_ok) p.in_ptr[N] = Etot, p.out_ptr[N] = Etot;
        }
    }
    if (!cap_ok || N <= 0) return;

    // ---- the weight fragments have arrived long ago: zero what lies outside W (dead K blocks, the k a shifted last block
    //      shares with its predecessor, columns >= NO) with dword masks -- pairs of k never straddle kstart (K % 4 == 0)
    u32x4 wf[PG_NB][PG_NT][NT];
#pragma unroll
    for (int s = 0; s < PG_NB; ++s) {
        const int kb = w + 8 * s;
        const int kstart = kb * 32, k = k0[s] + 8 * g;
        unsigned km[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) km[d] = (kb < nkb && k + 2 * d >= kstart) ? 0xffffffffu : 0u;
#pragma unroll
        for (int nt = 0; nt < PG_NT; ++nt) {
            const unsigned cm = n_base + 16 * nt + r < p.NO ? 0xffffffffu : 0u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const u32x4 v = wraw[s][nt][t];
                wf[s][nt][t] = (u32x4){v.x & km[0] & cm, v.y & km[1] & cm, v.z & km[2] & cm, v.w & km[3] & cm};
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    const int n_rg = (N + 15) / 16;
    for (int rg = pair; rg < n_rg; rg += n_pairs) {
        const int n0 = rg * 16;
        // ---- node -> dialogue of the 16 rows: dialogue t marks the rows of [n0, n0 + 16) it owns (first group: done above)
        if (rg != pair) {
            __syncthreads();
            for (int t = tid; t < p.B; t += 512) {
                const int lo = max(s_noff[t], n0), hi = min(s_noff[t + 1], n0 + 16);
                for (int n = lo; n < hi; ++n) s_dlg[n - n0] = t;
            }
            __syncthreads();
        }
        // ---- A fragments: feature row of node n0 + r is row b * T + position of the padded block
        bf16x8 af[PG_NB];
        f32x4 ax[NT > 1 ? PG_NB : 1][2];      // split modes: the raw fp32 values, expanded block by block in front of their products
        {
            const int nr = min(n0 + r, N - 1);
            const int b = s_dlg[nr - n0];
            const int64_t arow = ((int64_t)s_base[b] + (nr - s_noff[b])) * p.ldx;
#pragma unroll
            for (int s = 0; s < PG_NB; ++s) {
                if constexpr (NT == 1) {
                    const unsigned short* ar = reinterpret_cast<const unsigned short*>(p.X) + arow + k0[s] + 8 * g;
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(ar), hi = *reinterpret_cast<const bf16x4*>(ar + 4);
                    af[s] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                } else {
                    const float* ar = reinterpret_cast<const float*>(p.X) + arow + k0[s] + 8 * g;
                    ax[s][0] = *reinterpret_cast<const f32x4*>(ar), ax[s][1] = *reinterpret_cast<const f32x4*>(ar + 4);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the graph of these 16 nodes (one workgroup of the pair): thread (node i, slot); slots 0..15 walk the in-edges,
        //      16..31 the out-edges (windows wider than 16 take several trips).  The speakers of the node and of the first
        //      neighbour are requested here, next to the feature rows; offsets and stores follow behind the projection
        int sp = 0, s_nb = 0;
        if (half == 0) {
            const int i = tid >> 5, slot = tid & 31;
            const int nn = min(n0 + i, N - 1);
            const int b = s_dlg[nn - n0], noff = s_noff[b], Lb = s_noff[b + 1] - noff, pp = nn - noff;
            const int64_t* const spk = p.speakers + (p.desc ? (int64_t)s_base[b] * p.spk_st : (int64_t)b * p.spk_sb);
            const int back = slot < 16 ? p.wf : p.wp;
            const int first = min(max(0, pp - back) + (slot & 15), Lb - 1);
            sp = (int)spk[(int64_t)pp * p.spk_st];
            s_nb = (int)spk[(int64_t)first * p.spk_st];
        }
        // ---- 24 MFMAs, the 8 partial 16 x 64 tiles summed through LDS, bias, store
        f32x4 acc[PG_NT];
#pragma unroll
        for (int nt = 0; nt < PG_NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < PG_NB; ++s) {
            if constexpr (NT == 1) {
#pragma unroll
                for (int nt = 0; nt < PG_NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], __builtin_bit_cast(bf16x8, wf[s][nt][0]), acc[nt], 0, 0, 0);
            } else {
                u32x4 at[NT];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned tt[NT];
                    sp_split2<NT>(ax[s][d >> 1][2 * (d & 1)], ax[s][d >> 1][2 * (d & 1) + 1], tt);
#pragma unroll
                    for (int t = 0; t < NT; ++t) at[t][d] = tt[t];
                }
#pragma unroll
                for (int nt = 0; nt < PG_NT; ++nt) acc[nt] = sp_mfma<NT>(at, wf[s][nt], acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < PG_NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((w * PG_NT + nt) * 4 + i) * 64 + lane] = acc[nt][i];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < (PG_NT + 1) / 2; ++u) {
            const int e = tid + 512 * u, row = e / (16 * PG_NT), col = e % (16 * PG_NT);
            const int idx = (((col >> 4) * 4) + (row & 3)) * 64 + 16 * (row >> 2) + (col & 15);
            float sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) sum += red[ww * PG_NT * 4 * 64 + min(idx, PG_NT * 4 * 64 - 1)];
            const int mrow = n0 + row, ncol = n_base + col;
            if (e < 256 * PG_NT && mrow < N && ncol < p.NO) p.H0[(int64_t)mrow * p.ldh0 + ncol] = sum + p.bias[ncol];
        }
        if (half == 0) {
            const int i = tid >> 5, slot = tid & 31;
            const int n = n0 + i;
            const bool nv = n < N;
            const int nn = min(n, N - 1);
            const int b = s_dlg[nn - n0], noff = s_noff[b], Lb = s_noff[b + 1] - noff, pp = nn - noff, eoff = s_eoff[b];
            const int64_t* const spk = p.speakers + (p.desc ? (int64_t)s_base[b] * p.spk_st : (int64_t)b * p.spk_sb);
            if (slot < 16) {
                // in-edges of target pp: sources j in [pp - wf, pp + wp]  (canonical order: target-major, then source)
                const int lo = max(0, pp - p.wf), hi = min(Lb - 1, pp + p.wp);
                const int base = eoff + erc_window_prefix(pp, Lb, p.wf, p.wp);
                if (slot == 0 && nv) p.node_row[n] = s_base[b] + pp, p.node_spk[n] = sp, p.in_ptr[n] = base;
                for (int j = lo + slot; j <= hi; j += 16) {
                    const int sj = j == lo + slot ? s_nb : (int)spk[(int64_t)j * p.spk_st];
                    if (nv) {
                        const int e = base + (j - lo);
                        p.in_src[e] = noff + j;
                        p.in_typ[e] = 2 * (sj * p.S + sp) + (j < pp ? 0 : 1);
                    }
                }
            } else {
                // out-edges of source pp: targets k in [pp - wp, pp + wf]; out_eid = the edge's slot in the by-target CSR
                const int lo = max(0, pp - p.wp), hi = min(Lb - 1, pp + p.wf);
                const int base = eoff + erc_window_prefix(pp, Lb, p.wp, p.wf);
                if (slot == 16 && nv) p.out_ptr[n] = base;
                for (int k = lo + slot - 16; k <= hi; k += 16) {
                    const int sk = k == lo + slot - 16 ? s_nb : (int)spk[(int64_t)k * p.spk_st];
                    if (nv) {
                        const int e = base + (k - lo);
                        p.out_dst[e] = noff + k;
                        p.out_typ[e] = 2 * (sp * p.S + sk) + (pp < k ? 0 : 1);
                        p.out_eid[e] = eoff + erc_window_prefix(k, Lb, p.wf, p.wp) + (pp - max(0, k - p.wf));
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int erc_cogmen_project_graph_ok(int K, int n_out, int B, int ldx, int ldw) {
    return n_out > 0 && n_out <= 112 && K >= 32 && K <= 32 * 8 * PG_NB && K % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0 &&
           B > 0 && B <= PG_BMAX;
}

static int project_graph_launch(int terms, const void* X, int ldx, const void* W, int64_t w_plane, int ldw, const float* bias, float* H0,
                                int ldh0, int n_out, int K, const int64_t* lengths, const int64_t* speakers, int64_t spk_sb,
                                int64_t spk_st, int B, int T, int wp, int wf, int n_speakers, int n_cap, int e_cap,
                                int32_t* node_off, int32_t* node_row, int32_t* node_spk, int32_t* in_ptr, int32_t* in_src,
                                int32_t* in_typ, int32_t* out_ptr, int32_t* out_dst, int32_t* out_typ, int32_t* out_eid,
                                int32_t* counts, const int32_t* desc, void* stream) {
    ERC_REQUIRE(X && W && bias && H0 && (lengths || desc) && speakers && node_off && node_row && node_spk && in_ptr && in_src && in_typ &&
                    out_ptr && out_dst && out_typ && out_eid && counts,
                "cogmen_project_graph: null pointer");
    ERC_REQUIRE(erc_cogmen_project_graph_ok(K, n_out, B, ldx, ldw) && ((uintptr_t)X & (terms > 1 ? 15 : 7)) == 0 && ((uintptr_t)W & 7) == 0 &&
                    ldh0 >= n_out && T > 0 && n_speakers > 0 && n_cap > 0 && e_cap > 0,
                "cogmen_project_graph: unsupported sizes K=%d n_out=%d B=%d T=%d (erc_cogmen_project_graph_ok)", K, n_out, B, T);
    ERC_REQUIRE(wp >= -1 && wf >= -1, "cogmen_project_graph: window must be >= -1");
    ERC_REQUIRE(terms >= 1 && terms <= 3 && (terms == 1 || (w_plane > 0 && w_plane % 4 == 0)), "cogmen_project_graph: terms=%d w_plane=%lld",
                terms, (long long)w_plane);
    PgP p{X, (const unsigned short*)W, w_plane, bias, H0, ldx, ldw, ldh0, K, n_out, lengths, speakers, spk_sb, spk_st,
          B, T, wp < 0 ? T : wp, wf < 0 ? T : wf, n_speakers, n_cap, e_cap, node_off, node_row, node_spk, in_ptr, in_src, in_typ,
          out_ptr, out_dst, out_typ, out_eid, counts, desc};
    // 128 pairs of workgroups (bf16 mode: column tiles 0-3 | 4-6); 64 quads in the split modes (two tiles each: 96 / 144 registers of
    // resident weight fragments per lane; measured at config 2: two terms 13.8 us as quads, 16.0 as 80 triples of three tiles; three terms
    // 17.2 us as quads, 20.9 as 32 octets of one tile)
    if (terms == 1) hipLaunchKernelGGL((cogmen_project_graph_kernel<1, 4, 2>), dim3(256), dim3(512), 0, (hipStream_t)stream, p);
    else if (terms == 2) hipLaunchKernelGGL((cogmen_project_graph_kernel<2, 2, 4>), dim3(256), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((cogmen_project_graph_kernel<3, 2, 4>), dim3(256), dim3(512), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("cogmen_project_graph");
    return ERC_OK;
}

extern "C" int erc_cogmen_project_graph(const void* X, int ldx, const void* W, int ldw, const float* bias, float* H0, int ldh0,
                                        int n_out, int K, const int64_t* lengths, const int64_t* speakers, int64_t spk_sb,
                                        int64_t spk_st, int B, int T, int wp, int wf, int n_speakers, int n_cap, int e_cap,
                                        int32_t* node_off, int32_t* node_row, int32_t* node_spk, int32_t* in_ptr,
                                        int32_t* in_src, int32_t* in_typ, int32_t* out_ptr, int32_t* out_dst, int32_t* out_typ,
                                        int32_t* out_eid, int32_t* counts, const int32_t* desc, void* stream) {
    return project_graph_launch(1, X, ldx, W, 0, ldw, bias, H0, ldh0, n_out, K, lengths, speakers, spk_sb, spk_st, B, T, wp, wf,
                                n_speakers, n_cap, e_cap, node_off, node_row, node_spk, in_ptr, in_src, in_typ, out_ptr, out_dst,
                                out_typ, out_eid, counts, desc, stream);
}

// Split compute modes: X fp32 [rows, ldx] (16-byte aligned rows), W = `terms` (2 | 3) bf16 planes [n_out, ldw], w_plane elements apart
// (ErcShadowTab mode 0 with terms planes).  Everything else as erc_cogmen_project_graph.
extern "C" int erc_cogmen_project_graph_x(int terms, const float* X, int ldx, const void* W, int64_t w_plane, int ldw, const float* bias,
                                          float* H0, int ldh0, int n_out, int K, const int64_t* lengths, const int64_t* speakers,
                                          int64_t spk_sb, int64_t spk_st, int B, int T, int wp, int wf, int n_speakers, int n_cap,
                                          int e_cap, int32_t* node_off, int32_t* node_row, int32_t* node_spk, int32_t* in_ptr,
                                          int32_t* in_src, int32_t* in_typ, int32_t* out_ptr, int32_t* out_dst, int32_t* out_typ,
                                          int32_t* out_eid, int32_t* counts, const int32_t* desc, void* stream) {
    ERC_REQUIRE(terms == 2 || terms == 3, "cogmen_project_graph_x: terms = %d (2 or 3)", terms);
    return project_graph_launch(terms, X, ldx, W, w_plane, ldw, bias, H0, ldh0, n_out, K, lengths, speakers, spk_sb, spk_st, B, T, wp,
                                wf, n_speakers, n_cap, e_cap, node_off, node_row, node_spk, in_ptr, in_src, in_typ, out_ptr, out_dst,
                                out_typ, out_eid, counts, desc, stream);
}
