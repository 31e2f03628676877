// K1: device-side window-graph construction, one workgroup per dialogue.
//
// Replaces the reference's host-side python loops (one .item() per edge
// endpoint): track_mm/cogmen_utils.py:109-172, track_mm/dgcn_models.py:51-118.
// All offsets come from closed forms, so there is no
// sort, no atomics and no second pass: each dialogue writes its own slice of
// both CSRs and of the reference-shaped int64 edge list in canonical
// (target-major, then source) order.
#include "erc_common.h"

namespace {

__device__ __forceinline__ int window_prefix(int p, int L, int back, int fwd) { return erc_window_prefix(p, L, back, fwd); }

__global__ __launch_bounds__(256) void window_graph_kernel(
    const int64_t* __restrict__ lengths, const int64_t* __restrict__ speakers, int64_t spk_sb, int64_t spk_st,
    int B, int T, int wp, int wf, int S, int n_cap, int e_cap,
    int32_t* __restrict__ node_off, int32_t* __restrict__ node_row, int32_t* __restrict__ node_spk,
    int32_t* __restrict__ in_ptr, int32_t* __restrict__ in_src, int32_t* __restrict__ in_typ,
    int32_t* __restrict__ out_ptr, int32_t* __restrict__ out_dst, int32_t* __restrict__ out_typ,
    int32_t* __restrict__ out_eid, int64_t* __restrict__ edge_index, int64_t* __restrict__ edge_type,
    int32_t* __restrict__ counts) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ int red_n[4];
    __shared__ int red_e[4];
    extern __shared__ int s_spk[];  // speakers of this dialogue (T entries): one global round trip instead of one per edge

    const int64_t* spk = speakers + (int64_t)b * spk_sb;
    const int L = (int)lengths[b];
    for (int p = tid; p < min(L, T); p += 256) s_spk[p] = (int)spk[(int64_t)p * spk_st];
    // exclusive prefix of node / edge counts over the dialogues before b
    int acc_n = 0, acc_e = 0;
    for (int i = tid; i < b; i += 256) {
        int Li = (int)lengths[i];
        acc_n += Li;
        acc_e += window_prefix(Li, Li, wf, wp);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc_n += __shfl_xor(acc_n, o, 64), acc_e += __shfl_xor(acc_e, o, 64);
    if ((tid & 63) == 0) red_n[tid >> 6] = acc_n, red_e[tid >> 6] = acc_e;
    __syncthreads();
    const int noff = red_n[0] + red_n[1] + red_n[2] + red_n[3];
    const int eoff = red_e[0] + red_e[1] + red_e[2] + red_e[3];
    const int E_b = window_prefix(L, L, wf, wp);

    if (tid == 0) {
        node_off[b] = noff;
        if (b == B - 1) {
            node_off[B] = noff + L;
            counts[0] = noff + L;
            counts[1] = eoff + E_b;
            if (noff + L <= n_cap) {
                in_ptr[noff + L] = eoff + E_b;
                out_ptr[noff + L] = eoff + E_b;
            }
        }
    }
    if (noff + L > n_cap || eoff + E_b > e_cap || L > T) return;  // capacity guard (host checks counts)

    for (int p = tid; p < L; p += 256) {
        const int n = noff + p;
        const int sp = s_spk[p];
        node_row[n] = b * T + p;
        node_spk[n] = sp;

        // in-edges of target k = p: sources j in [p-wf, p+wp]
        {
            const int lo = max(0, p - wf), hi = min(L - 1, p + wp);
            const int base = eoff + window_prefix(p, L, wf, wp);
            in_ptr[n] = base;
            for (int j = lo; j <= hi; ++j) {
                const int e = base + (j - lo);
                const int sj = s_spk[j];
                const int ty = 2 * (sj * S + sp) + (j < p ? 0 : 1);
                in_src[e] = noff + j;
                in_typ[e] = ty;
                if (edge_index) {
                    edge_index[e] = noff + j;
                    edge_index[(int64_t)e_cap + e] = n;
                }
                if (edge_type) edge_type[e] = ty;
            }
        }
        // out-edges of source j = p: targets k in [p-wp, p+wf]
        {
            const int lo = max(0, p - wp), hi = min(L - 1, p + wf);
            const int base = eoff + window_prefix(p, L, wp, wf);
            out_ptr[n] = base;
            for (int k = lo; k <= hi; ++k) {
                const int e = base + (k - lo);
                const int sk = s_spk[k];
                out_dst[e] = noff + k;
                out_typ[e] = 2 * (sp * S + sk) + (p < k ? 0 : 1);
                out_eid[e] = eoff + window_prefix(k, L, wf, wp) + (p - max(0, k - wf));
            }
        }
    }
}

// test support: leave NaN bit patterns in the LDS of every CU, so that a kernel which reads LDS it did not write
// (e.g. the masked tail rows of an MFMA operand: 0 * NaN = NaN) fails its parity test on every run, not on some boxes
__global__ __launch_bounds__(256) void lds_poison_kernel(int words, int* sink) {
    extern __shared__ int lds_words[];
    for (int x = threadIdx.x; x < words; x += 256) lds_words[x] = -1;
    __syncthreads();
    if (sink && lds_words[(threadIdx.x * 97) % words] == 0) sink[0] = 1;    // keeps the stores alive
}

}  // namespace

extern "C" int erc_test_poison_lds(int32_t* sink, void* stream) {
    const int bytes = 160 * 1024;
    static bool raised = false;
    if (!raised) {
        ERC_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        bytes) == hipSuccess, "test_poison_lds: LDS limit");
        raised = true;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), bytes, (hipStream_t)stream, bytes / 4, sink);
    ERC_LAUNCH_CHECK("test_poison_lds");
    return ERC_OK;
}

extern "C" int erc_window_graph_build(const int64_t* lengths, const int64_t* speakers, int64_t spk_sb, int64_t spk_st,
                                      int B, int T, int wp, int wf, int n_speakers, int n_cap, int e_cap,
                                      int32_t* node_off, int32_t* node_row, int32_t* node_spk,
                                      int32_t* in_ptr, int32_t* in_src, int32_t* in_typ,
                                      int32_t* out_ptr, int32_t* out_dst, int32_t* out_typ, int32_t* out_eid,
                                      int64_t* edge_index, int64_t* edge_type, int32_t* counts, void* stream) {
    ERC_REQUIRE(B > 0 && T > 0 && n_speakers > 0, "window_graph_build: bad sizes B=%d T=%d S=%d", B, T, n_speakers);
    ERC_REQUIRE(lengths && speakers && node_off && node_row && node_spk && in_ptr && in_src && in_typ && out_ptr &&
                    out_dst && out_typ && out_eid && counts,
                "window_graph_build: null pointer");
    ERC_REQUIRE(wp >= -1 && wf >= -1, "window_graph_build: window must be >= -1");
    if (wp < 0) wp = T;  // -1 = unbounded past   (cogmen_utils.py:158-163)
    if (wf < 0) wf = T;  // -1 = unbounded future
    ERC_REQUIRE(T <= 12288, "window_graph_build: T=%d exceeds the LDS speaker stage", T);
    hipLaunchKernelGGL(window_graph_kernel, dim3(B), dim3(256), (size_t)T * sizeof(int), (hipStream_t)stream, lengths, speakers, spk_sb,
                       spk_st, B, T, wp, wf, n_speakers, n_cap, e_cap, node_off, node_row, node_spk, in_ptr, in_src,
                       in_typ, out_ptr, out_dst, out_typ, out_eid, edge_index, edge_type, counts);
    ERC_LAUNCH_CHECK("window_graph_build");
    return ERC_OK;
}
