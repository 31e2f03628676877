// (Bi)LSTM recurrence, hidden size 100 per direction -- the sequence-context encoders of DialogueGCN
// (SeqContext, packed: track_mm/dgcn_models.py:10-33) and MMGCN (text branch, unpacked over the padded
// length: track_mm/mmgcn.py:69,113-114).  torch.nn.LSTM semantics, gate order i|f|g|o.
//
// One workgroup per (dialogue, direction).  The input-side gate pre-activations GX = x W_ih^T + b_ih are one
// hoisted GEMM over all rows (both directions: 800 columns); the recurrent matrix W_hh [400,100] of the
// direction lives in REGISTERS for the whole scan (thread r owns gate row r: 100 VGPRs), h_{t-1} is
// broadcast from LDS, so a step costs ~100 LDS broadcasts + 100 FMAs per thread and touches HBM only for the
// GX row and the saved state.  The backward scan keeps the transposed slices in registers the same way and
// leaves all weight gradients to GEMMs over the saved gate gradients.
#include "erc_common.h"

namespace {

constexpr int H = 100;
constexpr int G4 = 400;
constexpr int NTH = 512;

struct LstmP {
    const float* GX; int ldgx;           // hoisted pre-activations, direction d at columns [400d, 400d+400)
    const float* W_hh;                   // [2][400,100]  (forward, reverse)
    const float* b_hh;                   // [2][400]
    const int64_t* lengths;              // [B] or null (= T for every dialogue: unpacked run)
    const int32_t* node_off;             // null: row(b,t) = b*sb + t*st ; else compact rows node_off[b] + t
    int64_t sb, st;
    int B, T;
    float* Hout; int ldh;                // outputs, direction d at columns [100d, 100d+100); rows as above
    float* Hdrop; int ldhd;              // optional copy with inverted dropout applied (input of the next layer)
    float drop_p; const uint64_t* rng;   // rng[0] = offset, rng[1] = seed
    uint64_t rng_stream;                 // distinguishes the layers' masks
    float* gates;                        // [rows,800] post-activation i|f|g|o per direction   (saved)
    float* Cst;                          // [rows,200] cell state                               (saved)
    float* Hprev;                        // [rows,200] h_{t-1} in scan order                     (saved)
    // backward only
    const float* dHout; int lddh;        // gradient wrt Hout (or wrt Hdrop when drop_p > 0)
    float* dGX;                          // [rows,800] gradient wrt the gate pre-activations (0 on padded rows)
};

__device__ __forceinline__ int64_t row_of(const LstmP& p, int b, int t) {
    return p.node_off ? (int64_t)p.node_off[b] + t : (int64_t)b * p.sb + (int64_t)t * p.st;
}
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(NTH) void lstm_fwd_kernel(LstmP p) {
    const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const int L = p.lengths ? (int)p.lengths[b] : p.T;
    __shared__ float s_h[H], s_c[H], s_g[G4];
    float w[H];
    if (tid < G4) {
        const float* src = p.W_hh + ((int64_t)d * G4 + tid) * H;
#pragma unroll
        for (int k = 0; k < H; ++k) w[k] = src[k];
    }
    const float bhh = tid < G4 ? p.b_hh[d * G4 + tid] : 0.f;
    if (tid < H) s_h[tid] = 0.f, s_c[tid] = 0.f;
    uint64_t roff = 0, rseed = 0;
    if (p.Hdrop && p.drop_p > 0.f) roff = p.rng[0], rseed = p.rng[1] ^ p.rng_stream;
    const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    __syncthreads();
    // the hoisted gate pre-activation of step s + 1 is requested while step s runs: its (L2) latency was the longest
    // item on the per-step critical path
    const int gcol = d * G4 + min(tid, G4 - 1);
    float gx_next = L > 0 ? p.GX[row_of(p, b, d == 0 ? 0 : L - 1) * p.ldgx + gcol] : 0.f;
    for (int s = 0; s < L; ++s) {
        const int t = d == 0 ? s : L - 1 - s;
        const int64_t row = row_of(p, b, t);
        const float gx = gx_next;
        {
            const int sn = min(s + 1, L - 1);
            gx_next = p.GX[row_of(p, b, d == 0 ? sn : L - 1 - sn) * p.ldgx + gcol];
        }
        if (tid < G4) {
            float a = gx + bhh;
#pragma unroll
            for (int k = 0; k < H; ++k) a += w[k] * s_h[k];
            const float act = (tid >= 2 * H && tid < 3 * H) ? tanhf(a) : sigm(a);
            s_g[tid] = act;
            p.gates[row * 2 * G4 + d * G4 + tid] = act;
        }
        __syncthreads();
        if (tid < H) {
            const float hprev = s_h[tid];
            const float c = s_g[H + tid] * s_c[tid] + s_g[tid] * s_g[2 * H + tid];
            const float h = s_g[3 * H + tid] * tanhf(c);
            s_c[tid] = c;
            s_h[tid] = h;
            p.Cst[row * 2 * H + d * H + tid] = c;
            p.Hprev[row * 2 * H + d * H + tid] = hprev;
            p.Hout[row * p.ldh + d * H + tid] = h;
            if (p.Hdrop) {
                float hd = h;
                if (p.drop_p > 0.f) {
                    const float u = erc_uniform(rseed, roff, (uint64_t)row * 2 * H + d * H + tid);
                    hd = u >= p.drop_p ? h * keep_scale : 0.f;
                }
                p.Hdrop[row * p.ldhd + d * H + tid] = hd;
            }
        }
        __syncthreads();
    }
    // padded positions: zero output (pad_packed_sequence) -- only meaningful for padded row addressing
    if (!p.node_off)
        for (int t = L; t < p.T; ++t) {
            const int64_t row = row_of(p, b, t);
            if (tid < H) {
                p.Hout[row * p.ldh + d * H + tid] = 0.f;
                if (p.Hdrop) p.Hdrop[row * p.ldhd + d * H + tid] = 0.f;
            }
        }
}

__global__ __launch_bounds__(NTH) void lstm_bwd_kernel(LstmP p) {
    const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const int L = p.lengths ? (int)p.lengths[b] : p.T;
    __shared__ float s_dh[H], s_dc[H], s_dp[G4], s_part[4][H];
    // thread (q,k), q = tid/100 < 4, holds W_hh[100q + j][k], j < 100: its share of (W_hh^T dpre)[k]
    const int q = tid / H, k = tid % H;
    float wt[H];
    if (tid < G4) {
        const float* src = p.W_hh + ((int64_t)d * G4 + q * H) * H + k;
#pragma unroll
        for (int j = 0; j < H; ++j) wt[j] = src[(int64_t)j * H];
    }
    if (tid < H) s_dh[tid] = 0.f, s_dc[tid] = 0.f;
    uint64_t roff = 0, rseed = 0;
    const bool dropped = p.drop_p > 0.f;
    if (dropped) roff = p.rng[0], rseed = p.rng[1] ^ p.rng_stream;
    const float keep_scale = dropped ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    __syncthreads();
    // operands of a step (upstream gradient, the four gates, cell state and previous cell state): requested one step
    // ahead, unconditionally (clamped rows), consumed from registers
    struct StepIn { float g, gi, gf, gg, go, c, cprev; };
    const int hc = min(tid, H - 1);
    auto fetch = [&](int s) {
        StepIn r;
        const int sc = max(s, 0);
        const int t = d == 0 ? sc : L - 1 - sc;
        const int64_t row = row_of(p, b, t);
        r.g = p.dHout[row * p.lddh + d * H + hc];
        const float* gt = p.gates + row * 2 * G4 + d * G4;
        r.gi = gt[hc], r.gf = gt[H + hc], r.gg = gt[2 * H + hc], r.go = gt[3 * H + hc];
        r.c = p.Cst[row * 2 * H + d * H + hc];
        const int sp = max(sc - 1, 0);
        const int tp = d == 0 ? sp : L - 1 - sp;
        r.cprev = p.Cst[row_of(p, b, tp) * 2 * H + d * H + hc] * (sc > 0 ? 1.f : 0.f);
        return r;
    };
    StepIn nxt = fetch(L - 1);
    for (int s = L - 1; s >= 0; --s) {
        const int t = d == 0 ? s : L - 1 - s;
        const int64_t row = row_of(p, b, t);
        const StepIn cur = nxt;
        nxt = fetch(s - 1);
        if (tid < H) {
            float g = cur.g;
            if (dropped) {
                const float u = erc_uniform(rseed, roff, (uint64_t)row * 2 * H + d * H + tid);
                g = u >= p.drop_p ? g * keep_scale : 0.f;
            }
            const float dh = g + s_dh[tid];
            const float gi = cur.gi, gf = cur.gf, gg = cur.gg, go = cur.go;
            const float c = cur.c;
            const float cprev = cur.cprev;
            const float tc = tanhf(c);
            const float dc = s_dc[tid] + dh * go * (1.f - tc * tc);
            const float dpi = dc * gg * gi * (1.f - gi);
            const float dpf = dc * cprev * gf * (1.f - gf);
            const float dpg = dc * gi * (1.f - gg * gg);
            const float dpo = dh * tc * go * (1.f - go);
            s_dc[tid] = dc * gf;
            s_dp[tid] = dpi; s_dp[H + tid] = dpf; s_dp[2 * H + tid] = dpg; s_dp[3 * H + tid] = dpo;
            float* o = p.dGX + row * 2 * G4 + d * G4;
            o[tid] = dpi; o[H + tid] = dpf; o[2 * H + tid] = dpg; o[3 * H + tid] = dpo;
        }
        __syncthreads();
        if (tid < G4) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < H; ++j) a += wt[j] * s_dp[q * H + j];
            s_part[q][k] = a;
        }
        __syncthreads();
        if (tid < H) s_dh[tid] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
        __syncthreads();
    }
    if (!p.node_off)
        for (int t = L; t < p.T; ++t) {
            const int64_t row = row_of(p, b, t);
            if (tid < G4) p.dGX[row * 2 * G4 + d * G4 + tid] = 0.f;
        }
}

}  // namespace

extern "C" int erc_lstm_scan_fwd(const float* GX, int ldgx, const float* W_hh, const float* b_hh,
                                 const int64_t* lengths, const int32_t* node_off, int64_t sb, int64_t st, int B, int T,
                                 float* Hout, int ldh, float* Hdrop, int ldhd, float drop_p, const uint64_t* rng_state,
                                 uint64_t rng_stream, float* gates, float* Cst, float* Hprev, void* stream) {
    ERC_REQUIRE(GX && W_hh && b_hh && Hout && gates && Cst && Hprev, "lstm_scan_fwd: null pointer");
    ERC_REQUIRE(B > 0 && T > 0 && ldgx >= 2 * G4 && ldh >= 2 * H, "lstm_scan_fwd: bad sizes B=%d T=%d", B, T);
    ERC_REQUIRE(!(Hdrop && drop_p > 0.f) || rng_state, "lstm_scan_fwd: dropout needs rng_state");
    LstmP p{};
    p.GX = GX; p.ldgx = ldgx; p.W_hh = W_hh; p.b_hh = b_hh; p.lengths = lengths; p.node_off = node_off;
    p.sb = sb; p.st = st; p.B = B; p.T = T; p.Hout = Hout; p.ldh = ldh; p.Hdrop = Hdrop; p.ldhd = ldhd;
    p.drop_p = drop_p; p.rng = rng_state; p.rng_stream = rng_stream; p.gates = gates; p.Cst = Cst; p.Hprev = Hprev;
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(B, 2), dim3(NTH), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("lstm_scan_fwd");
    return ERC_OK;
}

extern "C" int erc_lstm_scan_bwd(const float* W_hh, const int64_t* lengths, const int32_t* node_off, int64_t sb,
                                 int64_t st, int B, int T, const float* gates, const float* Cst, const float* dHout,
                                 int lddh, float drop_p, const uint64_t* rng_state, uint64_t rng_stream, float* dGX,
                                 void* stream) {
    ERC_REQUIRE(W_hh && gates && Cst && dHout && dGX, "lstm_scan_bwd: null pointer");
    ERC_REQUIRE(B > 0 && T > 0, "lstm_scan_bwd: bad sizes B=%d T=%d", B, T);
    ERC_REQUIRE(drop_p <= 0.f || rng_state, "lstm_scan_bwd: dropout needs rng_state");
    LstmP p{};
    p.W_hh = W_hh; p.lengths = lengths; p.node_off = node_off; p.sb = sb; p.st = st; p.B = B; p.T = T;
    p.gates = const_cast<float*>(gates); p.Cst = const_cast<float*>(Cst); p.dHout = dHout; p.lddh = lddh;
    p.drop_p = drop_p; p.rng = rng_state; p.rng_stream = rng_stream; p.dGX = dGX;
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(B, 2), dim3(NTH), 0, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("lstm_scan_bwd");
    return ERC_OK;
}
