// K6 (second generation): the DAG-ERC directed-acyclic recurrence (track_mm/dagerc.py:167-189,
// track_mm/dagerc_models.py:326-365) weight-stationary, with the batch of dialogues as the MFMA M dimension.
//
// The first-generation scan (round 1) gave every dialogue its own cluster of workgroups and re-streamed the
// 2.9 MB of recurrent fp32 weights from L2 at every one of the 110 steps: 41 GB of L2 traffic per training step for
// 11.6 MB of weights.  Here a GROUP of DG <= 16 dialogues is advanced together by P = 300 / EPC workgroups (one per CU).
// Workgroup c owns the hidden ELEMENTS E_c = [c EPC, (c+1) EPC) of every 300-vector of the recurrence and keeps, for its
// whole life, in registers (MFMA B operands, one VGPR per 4x16 block):
//   forward : rows {g 300 + e} of [W_hh(grus_c) ; W_ih(grus_p)] (6 per element), rows {e, 300 + e} of [Wr0 ; Wr1], w_k
//   backward: the same gate rows for the transposed product, and the COLUMNS E_c of Wr0 / Wr1.
// Per step the DG dialogues' vectors are the 16 rows of the MFMA A operand (v_mfma_f32_16x16x4_f32: exact fp32, a
// k-ordered fma chain), so one weight register serves all dialogues of the group.  What a step exchanges between the
// P workgroups are the two 300-vectors per dialogue that feed the two dependent products:
//   forward : all-gather M_i  -> gates -> GRU cells (slice-local) -> all-gather h_i -> R_i = Wr h_i, ks_i (slice-local)
//             -> attention over the window from the slice's own R history in LDS -> M_{i+1}[E_c]
//   backward: reduce-scatter of the partial dM_i = Wg[rows E_c]^T dgates (the transposed product yields partial FULL
//             vectors) -> all-gather dM_i -> Y_i = Wr[:, E_c]^T dM_i (slice-local), dalpha (300-dots against the saved R
//             rows) -> accumulations for the earlier steps (all slice-local or replicated).
//             The reference order "g_j = dH1_j + Wr^T dR_j" is re-associated to "g_j += alpha_ij Wr_sel^T dM_i" at the time
//             dM_i is known: the product then needs the all-gathered vector, not a second reduce-scatter.
// Exchange records are (value, tag) pairs in ONE 8-byte write-through (sc1) store, polled by the consumers themselves
// (MI355X_MICROARCH.md, hand-off price list: granules for latency); tag = epoch * 1024 + step + 1, the per-group epoch
// advances with every launch.  Every buffer is reused each step: a member can only overwrite a record of exchange X
// after it has consumed the OTHER exchange of the same step from every member, which each member publishes only
// after it has consumed X.
//
// Roles inside a workgroup (8 wavefronts): wavefronts 0..6 poll the all-gather records straight into MFMA A operands
// (K = 300 split 7 ways), multiply and leave 16x16 partial tiles in LDS; wavefront 7 owns everything elementwise (GRU
// cells, attention, publishing, saving what the backward needs) -- it never polls, so its bookkeeping stores never sit
// in front of a poll in a wavefront's in-order memory queue.  Two workgroup barriers per forward step, four per
// backward step.
//
// All G * P workgroups of a launch must be co-resident: the host sizes the launch from the occupancy query
// (erc_dag_rec_config) and splits the groups over several launches when the device cannot hold them; every poll is
// bounded and a timeout raises the error flag that makes the optimizer skip the step (erc_adam_step skip_flag).
#include "erc_common.h"

namespace {

constexpr int HID = 300;
constexpr int NTH = 512;        // threads per workgroup
constexpr int NMW = 7;          // matrix wavefronts
constexpr int EWW = 7;          // the elementwise wavefront
constexpr int KS = 75;          // k-steps of 4 over HID
constexpr int NQ = 11;          // k-steps per matrix wavefront (ceil(75 / 7))
constexpr int PST = 17;         // row pitch of a 16x16 partial tile in LDS
constexpr int XROW = 16;        // dialogue pitch of the all-gather records
constexpr int XG = HID * XROW;  // records per all-gather buffer and group
constexpr int NT19 = 19;        // 16-column tiles over HID
constexpr int TPW = 3;          // column tiles per matrix wavefront in the transposed product
constexpr int DMP = 308;        // row pitch of the gathered dM in LDS
constexpr int MAXDG = 16;
constexpr int SPIN_LIMIT = 4000000;
#ifndef POLL_SLEEP
#define POLL_SLEEP 2
#endif

typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ u64 ld64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_tag(u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_err(const int* e) { return __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void set_err(int* e) { __hip_atomic_store(e, ERC_HEALTH_RAISED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// re-poll one record until it carries `tag` (bounded; once any member has given up, everybody drains quickly)
__device__ __forceinline__ float settle(const u64* p, u64 v, unsigned tag, int& spins, int* err) {
    while ((unsigned)(v >> 32) != tag) {
        if (++spins > SPIN_LIMIT) {
            set_err(err);
            break;
        }
        if ((spins & 255) == 0 && ld_err(err)) break;
        __builtin_amdgcn_s_sleep(1);
        v = ld64(p);
    }
    return __builtin_bit_cast(float, (unsigned)v);
}

// The A operand of a matrix wavefront for an all-gathered [dialogue][300] vector set: lane l holds
// x[m = l & 15][k = 4 s + (l >> 4)] for its k-steps s = wave + 7 q; record (k, m) lives at k * 16 + m = 64 s + l.
template <int NWV, int NQT>
__device__ __forceinline__ void poll_operand(const u64* x, int wave, int lane, int ndlg, unsigned tag, float (&a)[NQT], int* err) {
    // What an exchange costs is set by the CONSUMER CU's own memory queue (MI355X_MICROARCH.md, handoff-1to1 by streaming
    // waves on the endpoint: 1.0 us idle, 2.3 - 2.8 us with 8 waves streaming): with all 77 record loads of the 7 matrix
    // wavefronts in flight and re-issued while waiting, an all-gather took 4.6 us.  So a wavefront first polls ONE
    // sentinel record (its first k-step) with a pause between polls, and requests the other ten only once that one has
    // arrived -- the producers publish within a fraction of a microsecond of each other, so those rarely need a retry.
    const bool mv = (lane & 15) < ndlg;
    int spins = 0;
    {
        u64 v0 = (u64)tag << 32;
        if (mv) {
            v0 = ld64(x + 64 * wave + lane);
            while ((unsigned)(v0 >> 32) != tag) {
                if (++spins > SPIN_LIMIT) {
                    set_err(err);
                    break;
                }
                if ((spins & 255) == 0 && ld_err(err)) break;
                __builtin_amdgcn_s_sleep(POLL_SLEEP);
                v0 = ld64(x + 64 * wave + lane);
            }
        }
        a[0] = __builtin_bit_cast(float, (unsigned)v0);
    }
    u64 v[NQT];
#pragma unroll
    for (int q = 1; q < NQT; ++q) {
        const int s = wave + NWV * q;
        v[q] = (u64)tag << 32;                     // no record for this lane / k-step: a zero operand
        if (mv && s < KS) v[q] = ld64(x + 64 * s + lane);
    }
#pragma unroll
    for (int q = 1; q < NQT; ++q) a[q] = settle(x + 64 * (wave + NWV * q) + lane, v[q], tag, spins, err);
}

__device__ __forceinline__ void store_tile(float* dst, const f32x4& acc, int lane) {
    // C/D layout of v_mfma_f32_16x16x4_f32: column = lane & 15, rows 4 (lane >> 4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(4 * (lane >> 4) + r) * PST + (lane & 15)] = acc[r];
}

__device__ __forceinline__ const float* gate_row(const float* W_hh_c, const float* W_ih_p, int gate, int e) {
    // row (gate, e) of the stacked sequential-side matrix [W_hh(grus_c) ; W_ih(grus_p)] ([1800, 300]): gates r, z, n of
    // cell C then of cell P
    return gate < 3 ? W_hh_c + (int64_t)(gate * HID + e) * HID : W_ih_p + (int64_t)((gate - 3) * HID + e) * HID;
}

constexpr int MAX_T = 1022;

// ----------------------------------------------------------------------------- meta
// speaker ids, DAG predecessor, valid-row map.  One workgroup per dialogue.
__global__ __launch_bounds__(256) void dag_meta_kernel(const float* __restrict__ onehot, const int64_t* __restrict__ ids,
                                                       int64_t sb, int64_t st, int S, const int64_t* __restrict__ lengths,
                                                       int B, int T, int32_t* __restrict__ spk, int32_t* __restrict__ pred,
                                                       int32_t* __restrict__ node_off, int32_t* __restrict__ node_row) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ int s_spk[1024];
    __shared__ int red[256];
    int acc = 0;
    for (int i = tid; i < b; i += 256) acc += (int)lengths[i];
    red[tid] = acc;
    for (int t = tid; t < T; t += 256) {
        int s = 0;
        if (onehot) {  // argmax of the one-hot row (first maximum, like torch.argmax)
            const float* row = onehot + (int64_t)b * sb + (int64_t)t * st;
            float best = row[0];
            for (int c = 1; c < S; ++c)
                if (row[c] > best) best = row[c], s = c;
        } else {
            s = (int)ids[(int64_t)b * sb + (int64_t)t * st];
        }
        s_spk[t] = s;
        spk[b * T + t] = s;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const int noff = red[0], L = (int)lengths[b];
    if (tid == 0) {
        node_off[b] = noff;
        if (b == B - 1) node_off[B] = noff + L;
    }
    for (int t = tid; t < T; t += 256) {
        int p = -1;
        for (int j = t - 1; j >= 0; --j)
            if (s_spk[j] == s_spk[t]) {
                p = j;
                break;
            }
        pred[b * T + t] = p;
        if (t < L) node_row[noff + t] = b * T + t;
    }
}


// ----------------------------------------------------------------------------------------------- forward
// The forward runs the LAYERS as a pipeline: layer l needs, at its step i, only h^{(l-1)}_{i+1} of the layer below, so
// the (up to) 4 layers of a launch run concurrently on their own workgroups, each about two steps behind the one below;
// 4 x 110 dependent steps become 110 + 6.  What used to be the hoisted GEMM of a layer (the input-side gates of cell C,
// the hidden-side gates of cell P and the attention's query score: [W_ih_c ; W_hh_p ; w_q] H_l) is a third stationary
// product of the layer's workgroups, fed by the records the layer below publishes anyway (its per-step all-gather of h
// is a RING over the steps, never overwritten inside a launch, so a consumer may lag); it is computed one step ahead
// and sits in the shadow of the layer's own all-gather of h.
constexpr int ML = 4;           // layers per launch (pipeline depth)
constexpr int FMW = 6;          // forward: matrix wavefronts 0..5, elementwise wavefronts 6 and 7 (8 dialogues each)
constexpr int FNQ = 13;         // k-steps per forward matrix wavefront (ceil(75 / 6))

struct FwdLayer {
    const float *Wh, *bh;                              // hoisted [1801(+1), 300] = W_ih(grus_c) ; W_hh(grus_p) ; w_q, and its biases [1801]
    const float *W_hh_c, *b_hh_c, *W_ih_p, *b_ih_p;    // sequential [900,300], [900]
    const float *Wr, *w_k;                             // [600,300] = Wr0 ; Wr1, [300]
    float *H1, *GI, *Mseq, *GH, *R, *ks, *alpha;       // outputs / saved for the backward
};

struct RecFwd {
    const float* H0; int ldh0;         // input of the launch's first layer [B*T, >= 300] (complete before the launch)
    FwdLayer ly[ML];
    int ldo, ldgi;                     // row pitch of H1 (>= 300) and of GI (>= 1801)
    const int32_t *pred, *spk;         // [B*T]
    int B, T, DG, g0, nl;              // dialogues per group, first group of this launch, layers of this launch
    u64 *xm;                           // [groups][ML][300][16]     all-gather records of M (reused every step)
    u64 *xh;                           // [groups][ML][T][300][16]  all-gather records of h, one set per step (ring)
    int *epoch, *err;                  // [groups], [1]
    u64* stamps;                       // diagnostic (NULL in production): [T][2][8] shader-clock stamps of workgroup 0
};

// diagnostic phase stamps of workgroup 0 (matrix wavefront 0 -> row 0, last elementwise wavefront -> row 1)
#define REC_STAMP(slot)                                                                                        \
    do {                                                                                                       \
        if (p.stamps && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7))                              \
            p.stamps[((int64_t)i * 2 + (wave == 7 ? 1 : 0)) * 8 + (slot)] = __builtin_amdgcn_s_memtime();      \
    } while (0)

template <int EPC>
__global__ __launch_bounds__(NTH) void dag_rec_fwd_kernel(RecFwd p) {
    constexpr int NTG = (6 * EPC + 15) / 16;            // 16-column tiles of this slice's sequential gate rows
    constexpr int NTH2 = (6 * EPC + 1 + 15) / 16;       // ... of its hoisted rows (+ the query-score column)
    constexpr int NSL = HID / EPC;
    static_assert(HID % EPC == 0 && 2 * EPC + 1 <= 16 && EPC <= 8, "slice width");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % NSL, l = (blockIdx.x / NSL) % p.nl, grp = p.g0 + blockIdx.x / (NSL * p.nl);
    const int DG = p.DG, T = p.T;
    const int b0 = grp * DG, ndlg = min(DG, p.B - b0);
    const FwdLayer& L = p.ly[l];
    float* part_g = smem;                               // [FMW][NTG][16][PST]
    float* part_h = part_g + FMW * NTG * 16 * PST;      // [FMW][NTH2][16][PST]
    float* part_r = part_h + FMW * NTH2 * 16 * PST;     // [FMW][16][PST]
    float* rhist = part_r + FMW * 16 * PST;             // [T][DG][2 EPC]  this slice's relation rows of the steps so far
    float* kshist = rhist + T * DG * 2 * EPC;           // [T][DG]         key scores
    int* s_spk = reinterpret_cast<int*>(kshist + T * DG);   // [DG][T]
    int* s_pred = s_spk + DG * T;
    for (int x = tid; x < ndlg * T; x += NTH) {
        s_spk[x] = p.spk[(int64_t)b0 * T + x];
        s_pred[x] = p.pred[(int64_t)b0 * T + x];
    }
    // the last layer's slice 0 advances the epoch when it is done: by then every workgroup of the group has read it
    const unsigned ep = (unsigned)p.epoch[grp] + 1u;
    u64* const xm = p.xm + ((int64_t)grp * ML + l) * XG;
    u64* const xh = p.xh + ((int64_t)grp * ML + l) * T * XG;            // this layer's ring
    const u64* const xlow = xh - (int64_t)T * XG;                       // the ring of the layer below (l > 0)

    // ---- stationary weights of the matrix wavefronts
    float wg[NTG][FNQ], wh[NTH2][FNQ], wr[FNQ];
#pragma unroll
    for (int q = 0; q < FNQ; ++q) {
        const int s = wave + FMW * q, kc = min(4 * s + (lane >> 4), HID - 1);
        const float kv = (wave < FMW && s < KS) ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            const int j = 16 * t + (lane & 15), jc = min(j, 6 * EPC - 1);
            wg[t][q] = gate_row(L.W_hh_c, L.W_ih_p, jc / EPC, c * EPC + jc % EPC)[kc] * (j < 6 * EPC ? kv : 0.f);
        }
#pragma unroll
        for (int t = 0; t < NTH2; ++t) {
            const int j = 16 * t + (lane & 15), jc = min(j, 6 * EPC);
            const int row = jc < 6 * EPC ? (jc / EPC) * HID + c * EPC + jc % EPC : 6 * HID;     // gate rows, then w_q
            wh[t][q] = L.Wh[(int64_t)row * HID + kc] * (j <= 6 * EPC ? kv : 0.f);
        }
        const int j = lane & 15, jc = min(j, 2 * EPC - 1);
        const float vr = L.Wr[(int64_t)((jc / EPC) * HID + c * EPC + jc % EPC) * HID + kc];
        wr[q] = (j < 2 * EPC ? vr : (j == 2 * EPC ? L.w_k[kc] : 0.f)) * kv;
    }
    // ---- items of the elementwise wavefronts: wavefront 6 + w owns dialogues [8 w, 8 w + 8), lane = el * 8 + (m & 7)
    const int el = lane >> 3, m = 8 * (wave - FMW) + (lane & 7), e = c * EPC + min(el, EPC - 1);
    const bool iv = wave >= FMW && el < EPC && m < ndlg;
    const int mc = iv ? m : 0;
    float bias[6], bhv[6], gi[6], mcur = 0.f, xcur = 0.f, qnext = 0.f;    // M_0 = 0 (dagerc.py:168-174)
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        bias[g] = g < 3 ? L.b_hh_c[g * HID + e] : L.b_ih_p[(g - 3) * HID + e];
        bhv[g] = L.bh[g * HID + e];
        gi[g] = 0.f;
    }
    const float bq = L.bh[6 * HID];
    __syncthreads();

    // the operand of the hoisted product of step i: row i of the layer below (a plain matrix for the first layer)
    auto lower_operand = [&](int i, float (&a)[FNQ]) {
        if (l == 0) {
            const bool mv = (lane & 15) < ndlg;
            const float* row = p.H0 + ((int64_t)(b0 + min(lane & 15, ndlg - 1)) * T + i) * p.ldh0;
#pragma unroll
            for (int q = 0; q < FNQ; ++q) {
                const int s = wave + FMW * q;
                a[q] = row[min(4 * s + (lane >> 4), HID - 1)] * ((mv && s < KS) ? 1.f : 0.f);
            }
        } else {
            poll_operand<FMW, FNQ>(xlow + (int64_t)i * XG, wave, lane, ndlg, ep * 1024u + (unsigned)i + 1u, a, p.err);
        }
    };
    auto hoisted_product = [&](int i) {               // matrix wavefronts: partial tiles of [W_ih_c ; W_hh_p ; w_q] h^{(l-1)}_i
        float a[FNQ];
        lower_operand(i, a);
        f32x4 acc[NTH2];
#pragma unroll
        for (int t = 0; t < NTH2; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < FNQ; ++q)
#pragma unroll
            for (int t = 0; t < NTH2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wh[t][q], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTH2; ++t) store_tile(part_h + (wave * NTH2 + t) * 16 * PST, acc[t], lane);
    };
    auto hoisted_reduce = [&](int i) {                // elementwise wavefronts: gates of step i (+ bias), query score, x
        if (!iv) return;
        const int64_t row = (int64_t)(b0 + m) * T + i;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int j = g * EPC + el;
            float s = bhv[g];
#pragma unroll
            for (int w = 0; w < FMW; ++w) s += part_h[((w * NTH2 + (j >> 4)) * 16 + m) * PST + (j & 15)];
            gi[g] = s;
            L.GI[row * p.ldgi + g * HID + e] = s;
        }
        {
            constexpr int j = 6 * EPC;
            float s = bq;
#pragma unroll
            for (int w = 0; w < FMW; ++w) s += part_h[((w * NTH2 + (j >> 4)) * 16 + m) * PST + (j & 15)];
            qnext = s;
            if (c == 0 && el == 0) L.GI[row * p.ldgi + 6 * HID] = s;
        }
        if (l == 0) {
            xcur = p.H0[row * p.ldh0 + e];
        } else {      // the record was consumed by the matrix wavefronts before the barrier: it is there
            const u64* rp = xlow + (int64_t)i * XG + e * XROW + m;
            int spins = 0;
            xcur = settle(rp, ld64(rp), ep * 1024u + (unsigned)i + 1u, spins, p.err);
        }
    };

    // ---- prologue: the hoisted product of step 0
    if (wave < FMW) hoisted_product(0);
    __syncthreads();
    hoisted_reduce(0);

    for (int i = 0; i < T; ++i) {
        const unsigned tag = ep * 1024u + (unsigned)i + 1u;
        REC_STAMP(0);
        if (wave < FMW) {
            // ---- gates = [W_hh_c ; W_ih_p][rows of E_c] . M_i
            f32x4 acc[NTG];
#pragma unroll
            for (int t = 0; t < NTG; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i > 0) {
                float a[FNQ];
                poll_operand<FMW, FNQ>(xm, wave, lane, ndlg, tag, a, p.err);
                REC_STAMP(1);
#pragma unroll
                for (int q = 0; q < FNQ; ++q)
#pragma unroll
                    for (int t = 0; t < NTG; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wg[t][q], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NTG; ++t) store_tile(part_g + (wave * NTG + t) * 16 * PST, acc[t], lane);
        }
        REC_STAMP(2);
        __syncthreads();
        REC_STAMP(3);
        if (wave < FMW) {
            // ---- the hoisted product of step i + 1 (its operand is normally there already: the layer below runs ahead) ...
            if (i + 1 < T) hoisted_product(i + 1);
            // ---- ... then R_i = Wr[rows of E_c] . h_i and ks_i = w_k . h_i
            float a[FNQ];
            poll_operand<FMW, FNQ>(xh + (int64_t)i * XG, wave, lane, ndlg, tag, a, p.err);
            REC_STAMP(4);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < FNQ; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wr[q], acc, 0, 0, 0);
            store_tile(part_r + wave * 16 * PST, acc, lane);
        } else if (iv) {
            // ---- the two GRU cells of this slice's elements (dagerc.py:175-186), h_i = C + P
            const int64_t row = (int64_t)(b0 + m) * T + i;
            float gh[6];
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const int j = g * EPC + el;
                float s = bias[g];
#pragma unroll
                for (int w = 0; w < FMW; ++w) s += part_g[((w * NTG + (j >> 4)) * 16 + m) * PST + (j & 15)];
                gh[g] = s;
            }
            float rr = sigm(gi[0] + gh[0]);
            float zz = sigm(gi[1] + gh[1]);
            float nn = tanhf(gi[2] + rr * gh[2]);
            const float cc = (1.f - zz) * nn + zz * mcur;               // cell C: x = H_l[i], h = M_i
            rr = sigm(gh[3] + gi[3]);
            zz = sigm(gh[4] + gi[4]);
            nn = tanhf(gh[5] + rr * gi[5]);
            const float pp = (1.f - zz) * nn + zz * xcur;               // cell P: x = M_i, h = H_l[i]
            const float h1 = cc + pp;
            st_tag(xh + (int64_t)i * XG + e * XROW + m, h1, tag);
            REC_STAMP(4);
            L.H1[row * p.ldo + e] = h1;
            L.Mseq[row * HID + e] = mcur;
#pragma unroll
            for (int g = 0; g < 6; ++g) L.GH[row * 6 * HID + g * HID + e] = gh[g];
        }
        REC_STAMP(5);
        __syncthreads();
        REC_STAMP(6);
        if (wave >= FMW) {
            if (iv) {
                const int64_t row = (int64_t)(b0 + m) * T + i;
                float r0 = 0.f, r1 = 0.f, kk = 0.f;
#pragma unroll
                for (int w = 0; w < FMW; ++w) {
                    const float* pr = part_r + (w * 16 + m) * PST;
                    r0 += pr[el], r1 += pr[EPC + el], kk += pr[2 * EPC];
                }
                rhist[(i * DG + m) * 2 * EPC + el] = r0;
                rhist[(i * DG + m) * 2 * EPC + EPC + el] = r1;
                L.R[row * 2 * HID + e] = r0;
                L.R[row * 2 * HID + HID + e] = r1;
                if (el == 0) {
                    kshist[i * DG + m] = kk;
                    if (c == 0) L.ks[row] = kk;
                }
            }
            if (i + 1 < T) {
                hoisted_reduce(i + 1);
                // ---- attention of step i + 1 over its DAG predecessors [max(pred, 0), i] (dagerc_models.py:326-365);
                //      every dialogue's items live in ONE wavefront, so the histories need no workgroup barrier
                if (iv) {
                    const int ii = i + 1;
                    const int pr = s_pred[mc * T + ii], lo = pr > 0 ? pr : 0, si = s_spk[mc * T + ii];
                    const float qs = qnext;
                    float mx = -INFINITY;
                    for (int j = lo; j <= i; ++j) mx = fmaxf(mx, qs + kshist[j * DG + m]);
                    float den = 0.f;
                    for (int j = lo; j <= i; ++j) den += expf(qs + kshist[j * DG + m] - mx);
                    const float inv = 1.0f / den;
                    float macc = 0.f;
                    const bool sv = c == 0 && el == 0;
                    float* arow = L.alpha + ((int64_t)(b0 + m) * T + ii) * T;
                    for (int j = lo; j <= i; ++j) {
                        const float al = expf(qs + kshist[j * DG + m] - mx) * inv;
                        macc += al * rhist[(j * DG + m) * 2 * EPC + (s_spk[m * T + j] == si ? 0 : EPC) + el];
                        if (sv) arow[j] = al;
                    }
                    mcur = macc;
                    st_tag(xm + e * XROW + m, macc, tag + 1u);
                }
            }
            REC_STAMP(7);
        }
    }
    if (l == p.nl - 1 && c == 0 && tid == 0) p.epoch[grp] = (int)ep;
}

// ----------------------------------------------------------------------------------------------- backward
// The backward pipelines the layers the same way, top layer first: layer l needs at its step i the complete gradient
// wrt h^{(l)}_i = (the head's part, a plain matrix) + (the input gradient of layer l + 1 at step i), and layer l + 1 knows
// its input gradient dH^{(l+1)}_i = z_p . g_i + [W_ih_c ; W_hh_p ; w_q]^T [dgates_i ; dqs_i] one step after it has processed
// step i.  That second term -- the former "dH_l += DGI Wh" GEMM -- is a second transposed product of the layer's
// workgroups; its partial vectors travel in the same reduce-scatter records as the partial dM, one step late.
// Per step: E1 (GRU cells backward) -> M1 (both transposed products, all 8 wavefronts) -> reduce-scatter -> E2 (dM of the
// slice; input gradient of the step before -> ring of the layer below / the dHall block) -> all-gather dM -> M2 (Y =
// Wr[:, E_c]^T dM; dalpha dots) -> E3 (softmax backward, accumulations for the earlier steps).
// The relation-weight gradient needs no accumulator in here: d[Wr0 ; Wr1] = sum_j dR_j h_j^T with dR_j = sum_i alpha_ij dM_i
// equals sum_i dM_i (sum_j alpha_ij h_j)^T, so the kernel saves dM and the caller multiplies it with the attention-weighted
// sums of the hidden states (erc_dag_attn_sums, a forward quantity).
constexpr int BTW = 3;          // column tiles of the transposed products per wavefront (all 8 wavefronts: 24 >= 19)

struct BwdLayer {
    const float* Hl;                                   // the layer's input [B*T, >= 300] (row pitch ldh)
    const float *GI, *GH, *Mseq, *R, *alpha;           // saved by the forward
    const float *Wh, *W_hh_c, *W_ih_p, *Wr, *w_k;
    const float* dHead;                                // [B*T, >= 300] (row pitch ldd): the part of dL/dh^{(l)} that does not
                                                       // come through the layer above inside this launch
    float *DGI, *DGH, *dM, *dks;                       // written: [B*T, lddgi], [B*T,1800], [B*T,300], [B*T]
};

struct RecBwd {
    BwdLayer ly[ML];                   // ly[0] = the LOWEST layer of the launch
    int ldh, ldgi, ldd, lddgi;
    float* dLow;                       // [B*T, >= 300] (row pitch ldd): += the input gradient of the launch's lowest layer
    int relu_low;                      // the lowest layer is layer 0 of the model: apply fc1's relu mask (Hl > 0) to dLow
    const int32_t *pred, *spk;
    int B, T, DG, g0, nl;
    u64 *xd;                           // [groups][ML][2][P consumers][P producers][ndlg][EPC] 16-byte records {partial dM, tag,
                                       // partial input gradient, tag}, two sets used by the parity of the step
    u64 *xm;                           // [groups][ML][300][16]     all-gather records of dM
    u64 *xu;                           // [groups][ML][T][300][16]  rings: complete input gradient of layer l per step
    int *epoch, *err;
    u64* stamps;
};

template <int EPC>
__global__ __launch_bounds__(NTH) void dag_rec_bwd_kernel(RecBwd p) {
    constexpr int NSL = HID / EPC;
    constexpr int NKG = (6 * EPC + 3) / 4;              // k-steps of the transposed sequential-side product (this slice's rows)
    constexpr int NKH = (6 * EPC + 1 + 3) / 4;          // ... of the hoisted-side product (+ the w_q row, slice 0 only)
    constexpr int VP = 4 * NKG + 1, VPH = 4 * NKH + 1;
    constexpr int SP = 16 * NT19 + 1;                   // row pitch of the staged partial vectors
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % NSL, l = (blockIdx.x / NSL) % p.nl, grp = p.g0 + blockIdx.x / (NSL * p.nl);
    const int DG = p.DG, T = p.T;
    const int b0 = grp * DG, ndlg = min(DG, p.B - b0);
    const BwdLayer& L = p.ly[l];
    const bool top = l == p.nl - 1, low = l == 0;
    float* vin = smem;                                  // [16][VP]      sequential-side gate gradients of this slice, step i
    float* vinh = vin + 16 * VP;                        // [2][16][VPH]  hoisted-side gate gradients + d(query score), by parity
    float* reda = vinh + 2 * 16 * VPH;                  // [NTH]
    float* redb = reda + NTH;                           // [NTH]
    float* part_y = redb + NTH;                         // [FMW][16][PST]
    float* dmfull = part_y + FMW * 16 * PST;            // [16][DMP]     the gathered dM_i
    float* stage = dmfull + 16 * DMP;                   // [2][16][SP]   both partial vectors of a step, [dialogue][k]
    float* gacc = stage + 2 * 16 * SP;                  // [T][DG][EPC]  sum_i alpha_ij (Wr_sel^T dM_i)[E_c]
    float* dks_s = gacc + T * DG * EPC;                 // [DG][T]
    float* dal = dks_s + DG * T;                        // [DG][T]       dalpha of the current step's window
    int* s_spk = reinterpret_cast<int*>(dal + DG * T);
    int* s_pred = s_spk + DG * T;
    for (int x = tid; x < ndlg * T; x += NTH) {
        s_spk[x] = p.spk[(int64_t)b0 * T + x];
        s_pred[x] = p.pred[(int64_t)b0 * T + x];
    }
    for (int x = tid; x < 16 * VP + 2 * 16 * VPH; x += NTH) vin[x] = 0.f;       // vin | vinh are contiguous
    for (int x = tid; x < T * DG * (EPC + 1); x += NTH) gacc[x] = 0.f;          // gacc | dks_s are contiguous
    const unsigned ep = (unsigned)p.epoch[grp] + 1u;
    const int64_t gl = (int64_t)grp * ML + l;
    u64* const xm = p.xm + gl * XG;
    u64* const xu = p.xu + gl * T * XG;                                  // this layer's ring (published when l > 0)
    const u64* const xup = xu + (int64_t)T * XG;                         // the ring of the layer above (read when !top)
    const int IT = ndlg * EPC;                                           // records per producer block
    // the reduce-scatter records of this (group, layer): two sets of NSL * NSL * IT 16-byte records; stores and loads
    // are 16-byte write-through / L1-bypassing buffer accesses (aux 16 = sc1), the descriptor built from uniform values
    const int64_t xd_set = (int64_t)NSL * NSL * DG * EPC * 2;            // u64 per set
    u64* const xdbase = p.xd + gl * 2 * xd_set;
    // (readfirstlane returns a SIGNED int: widen through uint32_t, or a low word >= 2^31 smears ones over the high word)
    const uint64_t xd_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)xdbase);
    const uint64_t xd_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)xdbase >> 32));
    const __amdgpu_buffer_rsrc_t xdr =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((xd_hi << 32) | xd_lo), 0, (int)(2 * xd_set * 8), 0x00020000);

    // ---- stationary weights: the transposed products use all 8 wavefronts (column tiles wave + 8 u), Y the first 6
    float wgT[BTW][NKG], whT[BTW][NKH], wrY[FNQ];
#pragma unroll
    for (int u = 0; u < BTW; ++u) {
        const int t = wave + 8 * u, k = 16 * t + (lane & 15), kc = min(k, HID - 1);
        const float kv = (t < NT19 && k < HID) ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < NKG; ++s) {
            const int n = 4 * s + (lane >> 4), nc = min(n, 6 * EPC - 1);
            wgT[u][s] = gate_row(L.W_hh_c, L.W_ih_p, nc / EPC, c * EPC + nc % EPC)[kc] * (n < 6 * EPC ? kv : 0.f);
        }
#pragma unroll
        for (int s = 0; s < NKH; ++s) {
            const int n = 4 * s + (lane >> 4), nc = min(n, 6 * EPC);
            const int row = nc < 6 * EPC ? (nc / EPC) * HID + c * EPC + nc % EPC : 6 * HID;
            const float on = n < 6 * EPC ? kv : ((n == 6 * EPC && c == 0) ? kv : 0.f);      // the w_q row counts once: slice 0
            whT[u][s] = L.Wh[(int64_t)row * HID + kc] * on;
        }
    }
#pragma unroll
    for (int q = 0; q < FNQ; ++q) {
        const int s = wave + FMW * q, ec = min(4 * s + (lane >> 4), HID - 1);
        const int j = lane & 15, jc = min(j, 2 * EPC - 1);
        const float v = L.Wr[(int64_t)((jc / EPC) * HID + ec) * HID + c * EPC + jc % EPC];
        wrY[q] = v * ((wave < FMW && s < KS && j < 2 * EPC) ? 1.f : 0.f);
    }
    // ---- items of the elementwise wavefronts (as in the forward): wavefront 6 + w owns dialogues [8 w, 8 w + 8)
    const int el = lane >> 3, m = 8 * (wave - FMW) + (lane & 7), e = c * EPC + min(el, EPC - 1);
    const bool iv = wave >= FMW && el < EPC && m < ndlg;
    const int mc = iv ? m : 0;
    const float wk_e = L.w_k[e];
    float dmdir = 0.f, dirh_prev = 0.f, dirh_cur = 0.f;   // z_c g (into M_i) ; z_p g (into H_l) of step i + 1 / of step i
    __syncthreads();

    // both transposed products of a step, all 8 wavefronts: partial dM_i (valid when do_a) and the partial input gradient
    // of step i + 1 (valid when do_b), published as two tagged records per (consumer slice, dialogue, element)
    auto transposed_products = [&](int i, bool do_a, bool do_b, unsigned tag, int set) {
        const float* vh = vinh + ((i + 1) & 1) * 16 * VPH;
        float a[NKG], b[NKH];
#pragma unroll
        for (int s = 0; s < NKG; ++s) a[s] = vin[(lane & 15) * VP + 4 * s + (lane >> 4)];
#pragma unroll
        for (int s = 0; s < NKH; ++s) b[s] = vh[(lane & 15) * VPH + 4 * s + (lane >> 4)];
#pragma unroll
        for (int u = 0; u < BTW; ++u) {
            f32x4 acca = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NKG; ++s) acca = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wgT[u][s], acca, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NKH; ++s) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], whT[u][s], accb, 0, 0, 0);
            const int t = wave + 8 * u;
            if (t < NT19) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    stage[(4 * (lane >> 4) + r) * SP + 16 * t + (lane & 15)] = acca[r];
                    stage[(16 + 4 * (lane >> 4) + r) * SP + 16 * t + (lane & 15)] = accb[r];
                }
            }
        }
        __syncthreads();
        // publish: record x = (consumer slice, dialogue, element) in consumer-major order -> every consumer's block of
        // this producer is one contiguous run of IT 16-byte records
        const unsigned ta = do_a ? tag : 0u, tb = do_b ? tag : 0u;
        for (int x = tid; x < NSL * IT; x += NTH) {
            const int cc = x / IT, o = x - cc * IT, mm = o / EPC, k = cc * EPC + (o - mm * EPC);
            u32x4 v;
            v.x = __builtin_bit_cast(unsigned, stage[mm * SP + k]), v.y = ta;
            v.z = __builtin_bit_cast(unsigned, stage[(16 + mm) * SP + k]), v.w = tb;
            __builtin_amdgcn_raw_buffer_store_b128(v, xdr, (int)((set * xd_set + ((int64_t)(cc * NSL + c) * IT + o) * 2) * 8), 0, 16);
        }
    };
    // every thread sums its share of the P partial blocks addressed to this slice, in producer order
    auto gather_partials = [&](bool do_a, bool do_b, unsigned tag, int set) {
        const int S = (NTH / IT) * IT, total = NSL * IT;
        const u64* blka = xdbase + set * xd_set + (int64_t)c * NSL * IT * 2;
        const u64* blkb = blka + 1;
        float suma = 0.f, sumb = 0.f;
        if (tid < S) {
            int spins = 0;
            if (tid < total) {   // sentinel: wait for this thread's first record before requesting the others (see poll_operand)
                const u64* sp = do_a ? blka + 2 * tid : blkb + 2 * tid;
                u64 v0 = ld64(sp);
                while ((unsigned)(v0 >> 32) != tag) {
                    if (++spins > SPIN_LIMIT) {
                        set_err(p.err);
                        break;
                    }
                    if ((spins & 255) == 0 && ld_err(p.err)) break;
                    __builtin_amdgcn_s_sleep(POLL_SLEEP);
                    v0 = ld64(sp);
                }
            }
            for (int x0 = tid; x0 < total; x0 += 4 * S) {
                u64 va[4], vb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int x = x0 + u * S;
                    va[u] = vb[u] = (u64)tag << 32;
                    if (x < total && do_a) va[u] = ld64(blka + 2 * x);
                    if (x < total && do_b) vb[u] = ld64(blkb + 2 * x);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int x = min(x0 + u * S, total - 1);
                    suma += settle(blka + 2 * x, va[u], tag, spins, p.err);
                    sumb += settle(blkb + 2 * x, vb[u], tag, spins, p.err);
                }
            }
        }
        reda[tid] = suma, redb[tid] = sumb;
    };
    // the complete input gradient of step j for this slice: to the ring of the layer below, or (lowest layer of the
    // launch) added to the dHall block, through fc1's relu mask on layer 0 of the model
    auto emit_input_gradient = [&](int j, unsigned tagj) {
        if (!iv) return;
        const int S = (NTH / IT) * IT;
        float v = dirh_prev;
        for (int x = m * EPC + el; x < S; x += IT) v += redb[x];
        const int64_t row = (int64_t)(b0 + m) * T + j;
        if (!low) {
            st_tag(xu + (int64_t)j * XG + e * XROW + m, v, tagj);
        } else {
            float* dst = p.dLow + row * p.ldd + e;
            const float keep = (!p.relu_low || L.Hl[row * p.ldh + e] > 0.f) ? 1.f : 0.f;
            *dst = (*dst + v) * keep;
        }
    };

    for (int i = T - 1; i >= 0; --i) {
        const unsigned tag = ep * 1024u + (unsigned)i + 1u;
        REC_STAMP(0);
        // ---- E1: total gradient wrt h_i for this slice, GRU cells backward (elementwise)
        if (iv) {
            const int64_t row = (int64_t)(b0 + m) * T + i;
            const float* gi = L.GI + row * p.ldgi;
            const float* gh = L.GH + row * 6 * HID;
            float giv[6], ghv[6];
#pragma unroll
            for (int g = 0; g < 6; ++g) giv[g] = gi[g * HID + e], ghv[g] = gh[g * HID + e];
            const float mi = L.Mseq[row * HID + e], xi = L.Hl[row * p.ldh + e];
            float g = L.dHead[row * p.ldd + e] + gacc[(i * DG + m) * EPC + el] + wk_e * dks_s[m * T + i];
            if (!top) {      // + the input gradient of the layer above at this step (it runs about two steps ahead)
                const u64* rp = xup + (int64_t)i * XG + e * XROW + m;
                int spins = 0;
                u64 v0 = ld64(rp);
                while ((unsigned)(v0 >> 32) != tag) {
                    if (++spins > SPIN_LIMIT) {
                        set_err(p.err);
                        break;
                    }
                    if ((spins & 255) == 0 && ld_err(p.err)) break;
                    __builtin_amdgcn_s_sleep(POLL_SLEEP);
                    v0 = ld64(rp);
                }
                g += __builtin_bit_cast(float, (unsigned)v0);
            }
            float* dgi = L.DGI + row * p.lddgi;
            float* dgh = L.DGH + row * 6 * HID;
            float* vrow = vin + m * VP;
            float* vhrow = vinh + ((i & 1) * 16 + m) * VPH;
            {   // cell C: x = H_l[i] (hoisted side), h = M_i (sequential side)
                const float rr = sigm(giv[0] + ghv[0]), zz = sigm(giv[1] + ghv[1]);
                const float nn = tanhf(giv[2] + rr * ghv[2]);
                const float dn = g * (1.f - zz) * (1.f - nn * nn);
                const float dz = g * (mi - nn) * zz * (1.f - zz);
                const float dr = dn * ghv[2] * rr * (1.f - rr);
                dgi[e] = dr, dgi[HID + e] = dz, dgi[2 * HID + e] = dn;
                dgh[e] = dr, dgh[HID + e] = dz, dgh[2 * HID + e] = dn * rr;
                vrow[el] = dr, vrow[EPC + el] = dz, vrow[2 * EPC + el] = dn * rr;
                vhrow[el] = dr, vhrow[EPC + el] = dz, vhrow[2 * EPC + el] = dn;
                dmdir = g * zz;                                        // direct path into M_i
            }
            {   // cell P: x = M_i (sequential side), h = H_l[i] (hoisted side)
                const float rr = sigm(ghv[3] + giv[3]), zz = sigm(ghv[4] + giv[4]);
                const float nn = tanhf(ghv[5] + rr * giv[5]);
                const float dn = g * (1.f - zz) * (1.f - nn * nn);
                const float dz = g * (xi - nn) * zz * (1.f - zz);
                const float dr = dn * giv[5] * rr * (1.f - rr);
                dgh[3 * HID + e] = dr, dgh[4 * HID + e] = dz, dgh[5 * HID + e] = dn;
                dgi[3 * HID + e] = dr, dgi[4 * HID + e] = dz, dgi[5 * HID + e] = dn * rr;
                vrow[3 * EPC + el] = dr, vrow[4 * EPC + el] = dz, vrow[5 * EPC + el] = dn;
                vhrow[3 * EPC + el] = dr, vhrow[4 * EPC + el] = dz, vhrow[5 * EPC + el] = dn * rr;
                dirh_cur = g * zz;                                     // direct path into H_l[i]
            }
            if (i == 0 && el == 0) {      // step 0 has no attention: d(query score) = 0
                vhrow[6 * EPC] = 0.f;
                if (c == 0) dgi[6 * HID] = 0.f;
            }
        }
        REC_STAMP(1);
        __syncthreads();
        // ---- M1: partial dM_i = Wg[rows of E_c]^T dgates_i and the partial input gradient of step i + 1
        const bool do_a = i > 0, do_b = i + 1 < T;       // M_0 = 0 has no producers; step T - 1 has no later step
        transposed_products(i, do_a, do_b, tag, i & 1);
        REC_STAMP(2);
        gather_partials(do_a, do_b, tag, i & 1);
        REC_STAMP(3);
        __syncthreads();
        // ---- the window's (dialogue, j) pairs, COMPACTED: lane m < 16 of every wavefront holds the window size of dialogue m, a
        //      wavefront scan gives the pair offsets and a ballot maps pair d back to its dialogue -- sum_m n_m pairs instead of
        //      ndlg * max_m n_m (a third of them at two speakers: one dependent batch of row loads instead of three).
        const int nm = lane < ndlg ? i - max(s_pred[min(lane, ndlg - 1) * T + i], 0) : 0;
        int incl = nm;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        const int excl = incl - nm;
        const int npair = __shfl(incl, 15, 64);
        // pair d -> (dialogue, offset in its window); recomputed where needed instead of carried across the all-gather
        auto pair_of = [&](int d0, int u, int& mm, int& jj) {
            const int d = max(min(d0 + u, npair - 1), 0);
            const unsigned long long hit = __ballot(lane < 16 && excl <= d && d < incl);
            mm = min(__builtin_amdgcn_readfirstlane(__builtin_ctzll(hit | (1ull << 63))), ndlg - 1);
            jj = max(d - __shfl(excl, mm, 64), 0);
        };
        auto load_batch = [&](int ii, int d0, float (&vv)[4][5]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int mm, jj;
                pair_of(d0, u, mm, jj);
                const int pr = s_pred[mm * T + ii], lo = pr > 0 ? pr : 0;
                const int j = min(lo + jj, T - 1);
                const float* v = L.R + ((int64_t)(b0 + mm) * T + j) * 2 * HID + (s_spk[mm * T + j] == s_spk[mm * T + ii] ? 0 : HID);
#pragma unroll
                for (int r = 0; r < 5; ++r) vv[u][r] = v[min(lane + 64 * r, HID - 1)];
            }
        };
        auto dot_batch = [&](int d0, float (&vv)[4][5]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int mm, jj;
                pair_of(d0, u, mm, jj);
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const int k = lane + 64 * r;
                    s += vv[u][r] * dmfull[mm * DMP + min(k, HID - 1)] * (k < HID ? 1.f : 0.f);
                }
                s = wave_sum(s);
                if (lane == 0 && d0 + u < npair) dal[mm * T + jj] = s;
            }
        };
        // (Requesting the first batch of rows HERE, in front of the all-gather they do not depend on, was measured and lost:
        //  the 20 registers they occupy across the poll spill, and a spilled load result is a wait -- 39.9 k cycles per step
        //  against 38.3 k.)
        // ---- E2 / M2: dM_i of this slice -> all-gather; the input gradient of step i + 1 goes out; Y_i = Wr[:, E_c]^T dM_i
        if (wave >= FMW) {
            if (iv) {
                float dm = 0.f;                                         // M_0 = 0: no dM for step 0
                if (do_a) {
                    const int S = (NTH / IT) * IT;
                    dm = dmdir;
                    for (int x = m * EPC + el; x < S; x += IT) dm += reda[x];
                    st_tag(xm + e * XROW + m, dm, tag);
                }
                L.dM[((int64_t)(b0 + m) * T + i) * HID + e] = dm;
            }
            if (do_b) emit_input_gradient(i + 1, tag + 1u);
            dirh_prev = dirh_cur;
        } else if (do_a) {
            float a[FNQ];
            poll_operand<FMW, FNQ>(xm, wave, lane, ndlg, tag, a, p.err);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < FNQ; ++q) {
                const int s = wave + FMW * q;
                if (s < KS) dmfull[(lane & 15) * DMP + 4 * s + (lane >> 4)] = a[q];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wrY[q], acc, 0, 0, 0);
            }
            store_tile(part_y + wave * 16 * PST, acc, lane);
        }
        REC_STAMP(4);
        if (i == 0) break;
        __syncthreads();
        REC_STAMP(5);
        // ---- dalpha_ij = dM_i . V_j over the window (V_j = the relation row of j that step i read): all 8 wavefronts, four
        //      compacted (dialogue, j) pairs per wavefront and batch
        for (int d0 = 4 * wave; d0 < npair; d0 += 32) {
            float vv[4][5];
            load_batch(i, d0, vv);
            dot_batch(d0, vv);
        }
        REC_STAMP(6);
        __syncthreads();
        // ---- E3: softmax backward, accumulations for the earlier steps
        if (iv) {
            float y0 = 0.f, y1 = 0.f;
#pragma unroll
            for (int w = 0; w < FMW; ++w) {
                const float* py = part_y + (w * 16 + m) * PST;
                y0 += py[el], y1 += py[EPC + el];
            }
            const int pr = s_pred[m * T + i], lo = pr > 0 ? pr : 0, si = s_spk[m * T + i], n = i - lo;
            const float* arow = L.alpha + ((int64_t)(b0 + m) * T + i) * T + lo;
            float t = 0.f;
            for (int jj = 0; jj < n; ++jj) t += arow[jj] * dal[m * T + jj];
            float dq = 0.f;
            for (int jj = 0; jj < n; ++jj) {
                const int j = lo + jj;
                const float al = arow[jj];
                const float ds = al * (dal[m * T + jj] - t);
                dq += ds;
                if (el == 0) dks_s[m * T + j] += ds;
                gacc[(j * DG + m) * EPC + el] += al * (s_spk[m * T + j] == si ? y0 : y1);
            }
            if (el == 0) {
                vinh[((i & 1) * 16 + m) * VPH + 6 * EPC] = dq;          // d(query score) joins the hoisted-side row of step i
                if (c == 0) L.DGI[((int64_t)(b0 + m) * T + i) * p.lddgi + 6 * HID] = dq;
            }
        }
        REC_STAMP(7);
    }
    // ---- flush: the input gradient of step 0 (one more round of the second transposed product).  It travels in the
    //      record set of parity 1: round 0 published set 0 without a dM all-gather behind it (M_0 has none), so a fast
    //      member could overwrite a round-0 record that a slow one has not read yet; set 1 was last used in round 1,
    //      which the all-gather of dM_1 closed for every member.
    __syncthreads();
    {
        const unsigned tagf = ep * 1024u + (unsigned)T + 2u;
        transposed_products(-1, false, true, tagf, 1);
        gather_partials(false, true, tagf, 1);
        __syncthreads();
        if (wave >= FMW) emit_input_gradient(0, ep * 1024u + 1u);
    }
    if (iv && el == 0 && c == 0)
        for (int j = 0; j < T; ++j) L.dks[(int64_t)(b0 + m) * T + j] = dks_s[m * T + j];
    if (low && c == 0 && tid == 0) p.epoch[grp] = (int)ep;
}

// dR_j = sum_i alpha_ij dM_i never has to exist: d[Wr0 ; Wr1] = sum_i dM_i (A_i)^T with the attention-weighted sums
// A_i[sel] = sum_{j in window(i), speaker relation sel} alpha_ij h_j -- a forward quantity, one workgroup per row.
__global__ __launch_bounds__(64) void dag_attn_sums_kernel(const float* __restrict__ alpha, const float* __restrict__ H1, int ldo,
                                                           const int32_t* __restrict__ pred, const int32_t* __restrict__ spk,
                                                           int T, float* __restrict__ A) {
    const int64_t row = blockIdx.x;
    const int i = (int)(row % T), lane = threadIdx.x;
    const int64_t base = row - i;
    const int pr = pred[row], lo = pr > 0 ? pr : 0, si = spk[row];
    float acc[2][5];
#pragma unroll
    for (int r = 0; r < 5; ++r) acc[0][r] = acc[1][r] = 0.f;
    for (int j = lo; j < i; ++j) {
        const float al = alpha[row * T + j];
        const float a0 = spk[base + j] == si ? al : 0.f, a1 = al - a0;
        const float* h = H1 + (base + j) * ldo;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const float hv = h[min(lane + 64 * r, HID - 1)];
            acc[0][r] += a0 * hv, acc[1][r] += a1 * hv;
        }
    }
#pragma unroll
    for (int r = 0; r < 5; ++r)
        if (lane + 64 * r < HID) A[row * 2 * HID + lane + 64 * r] = acc[0][r], A[row * 2 * HID + HID + lane + 64 * r] = acc[1][r];
}

// ----------------------------------------------------------------------------------------------- host side
int lds_fwd(int epc, int dg, int T) {
    const int ntg = (6 * epc + 15) / 16, nth2 = (6 * epc + 1 + 15) / 16;
    return 4 * (FMW * (ntg + nth2 + 1) * 16 * PST + T * dg * 2 * epc + T * dg + 2 * dg * T);
}
int lds_bwd(int epc, int dg, int T) {
    const int nkg = (6 * epc + 3) / 4, nkh = (6 * epc + 1 + 3) / 4;
    return 4 * (16 * (4 * nkg + 1) + 2 * 16 * (4 * nkh + 1) + 2 * NTH + FMW * 16 * PST + 16 * DMP + 2 * 16 * (16 * NT19 + 1) +
                T * dg * (epc + 1) + dg * T + 2 * dg * T);
}

int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

// The kernels use more than 64 KB of dynamic LDS: the per-kernel limit must be raised first, and only ever RAISED (the
// configuration search probes candidates with smaller needs after the one it finally picks; a limit left at the last
// candidate's value made launches of the chosen configuration run with too little LDS).
template <typename K>
bool ensure_lds(K kernel, int lds) {
    // keyed by the kernel's address: every EPC instantiation has the same function-pointer TYPE
    static const void* known[16];
    static int granted[16];
    static int n_known = 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    int slot = -1;
    for (int i = 0; i < n_known; ++i)
        if (known[i] == key) slot = i;
    if (slot < 0) {
        if (n_known == 16) return false;
        slot = n_known++;
        known[slot] = key, granted[slot] = 64 * 1024;
    }
    if (lds <= granted[slot]) return true;
    if (hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return false;
    granted[slot] = lds;
    return true;
}

// workgroups of a kernel that the device holds at once (one per CU is what the exchange latency is tuned for; never more
// than the occupancy query admits)
template <typename K>
int capacity(K kernel, int lds) {
    const int cus = device_cus();
    if (cus < 0) return -1;
    if (lds > 160 * 1024) return 0;
    if (!ensure_lds(kernel, lds)) return -1;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, NTH, lds) != hipSuccess) return -1;
    return cus * (n < 1 ? n : 1);
}

int capacity_of(int dir, int epc, int dg, int T) {
    const int lds = dir ? lds_bwd(epc, dg, T) : lds_fwd(epc, dg, T);
    switch (epc) {
        case 2: return dir ? capacity(dag_rec_bwd_kernel<2>, lds) : capacity(dag_rec_fwd_kernel<2>, lds);
        case 4: return dir ? capacity(dag_rec_bwd_kernel<4>, lds) : capacity(dag_rec_fwd_kernel<4>, lds);
        case 5: return dir ? capacity(dag_rec_bwd_kernel<5>, lds) : capacity(dag_rec_fwd_kernel<5>, lds);
        default: return 0;
    }
}

bool cfg_ok(const int* cfg) {
    return cfg && (cfg[0] == 2 || cfg[0] == 4 || cfg[0] == 5) && cfg[1] >= 1 && cfg[1] <= MAXDG && cfg[2] >= 1 && cfg[3] >= 1 && cfg[3] <= ML;
}

}  // namespace

extern "C" int erc_dag_meta(const float* speaker_onehot, const int64_t* speaker_ids, int64_t spk_sb, int64_t spk_st,
                            int n_speakers, const int64_t* lengths, int B, int T, int32_t* spk, int32_t* pred,
                            int32_t* node_off, int32_t* node_row, void* stream) {
    ERC_REQUIRE((speaker_onehot != nullptr) != (speaker_ids != nullptr), "dag_meta: give one-hot OR ids");
    ERC_REQUIRE(lengths && spk && pred && node_off && node_row, "dag_meta: null pointer");
    ERC_REQUIRE(B > 0 && T > 0 && T <= MAX_T && n_speakers > 0, "dag_meta: B=%d T=%d (T <= %d)", B, T, MAX_T);
    hipLaunchKernelGGL(dag_meta_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, speaker_onehot, speaker_ids, spk_sb,
                       spk_st, n_speakers, lengths, B, T, spk, pred, node_off, node_row);
    ERC_LAUNCH_CHECK("dag_meta");
    return ERC_OK;
}


// cfg[4] = {elements per workgroup, dialogues per group, groups per launch, layers per launch}
extern "C" int erc_dag_rec_config(int dir, int B, int T, int n_layers, int epc_hint, int dg_hint, int lpl_hint, int* cfg) {
    ERC_REQUIRE(B > 0 && T > 0 && T < 1023 && n_layers > 0 && cfg, "dag_rec_config: B=%d T=%d layers=%d", B, T, n_layers);
    const int cand[3] = {5, 4, 2};
    double best = 1e30;
    cfg[0] = 0;
    for (int ci = 0; ci < 3; ++ci) {
        const int epc = cand[ci], P = HID / epc;
        if (epc_hint > 0 && epc != epc_hint) continue;
        for (int dg = (dg_hint > 0 ? dg_hint : 1); dg <= (dg_hint > 0 ? dg_hint : MAXDG); ++dg) {
            const int cap = capacity_of(dir, epc, dg, T);
            ERC_REQUIRE(cap >= 0, "dag_rec_config: device query failed (no GPU?)");
            const int groups = erc_cdiv(B, dg);
            const int lmax = n_layers < ML ? n_layers : ML;                  // both directions pipeline the layers
            for (int lpl = (lpl_hint > 0 ? lpl_hint : 1); lpl <= (lpl_hint > 0 ? lpl_hint : lmax); ++lpl) {
                if (lpl > lmax) break;
                int gpl = cap / (lpl * P);
                if (gpl < 1) continue;
                if (gpl > groups) gpl = groups;
                // dependent steps of all launches; a mild preference for fewer dialogues per group (shorter elementwise
                // phases, smaller records) and for more elements per workgroup (fewer members per exchange)
                const double steps = (double)erc_cdiv(groups, gpl) * erc_cdiv(n_layers, lpl) * (T + 2 * (lpl - 1));
                const double cost = steps * (1.0 + 0.01 * dg + 0.02 * ci);
                if (cost < best) best = cost, cfg[0] = epc, cfg[1] = dg, cfg[2] = gpl, cfg[3] = lpl;
            }
        }
    }
    ERC_REQUIRE(cfg[0], "dag_rec_config: no configuration fits this device (B=%d T=%d: LDS per workgroup or CU count)", B, T);
    return ERC_OK;
}

static u64* g_stamps = nullptr;

// diagnostic: the next launches record shader-clock stamps of workgroup 0 into stamps[T][2][8] (NULL switches it off)
extern "C" int erc_dag_rec_set_stamps(uint64_t* stamps) {
    g_stamps = reinterpret_cast<u64*>(stamps);
    return ERC_OK;
}

// scratch (bytes).  forward: all-gather records of M [groups][4][300][16] and the per-step rings of h
// [groups][4][T][300][16]; backward: all-gather records of dM and the per-step rings of the input gradients (same
// shapes), two sets of reduce-scatter records [groups][4][P][P][dg][epc] (16 bytes each)
extern "C" int64_t erc_dag_rec_scratch_bytes(int dir, int B, int T, const int* cfg) {
    if (B <= 0 || T <= 0 || !cfg_ok(cfg)) return -1;
    const int64_t groups = erc_cdiv(B, cfg[1]), P = HID / cfg[0];
    const int64_t rings = 8 * (groups * ML * XG * (1 + (int64_t)T));
    return dir ? rings + 8 * 4 * groups * ML * P * P * cfg[1] * cfg[0] : rings;      // two sets of 16-byte records
}

#define REC_DISPATCH(KERNEL, ARGS, LDS, WGS)                                                                    \
    switch (cfg[0]) {                                                                                           \
        case 2: ensure_lds(KERNEL<2>, LDS); hipLaunchKernelGGL(KERNEL<2>, dim3((WGS) * (HID / 2)), dim3(NTH), LDS, st, ARGS); break; \
        case 4: ensure_lds(KERNEL<4>, LDS); hipLaunchKernelGGL(KERNEL<4>, dim3((WGS) * (HID / 4)), dim3(NTH), LDS, st, ARGS); break; \
        case 5: ensure_lds(KERNEL<5>, LDS); hipLaunchKernelGGL(KERNEL<5>, dim3((WGS) * (HID / 5)), dim3(NTH), LDS, st, ARGS); break; \
    }

extern "C" int erc_dag_rec_fwd(const float* H0, int ldh0, int n_layers, const float* const* Wh, const float* const* bh,
                               const float* const* W_hh_c, const float* const* b_hh_c, const float* const* W_ih_p,
                               const float* const* b_ih_p, const float* const* Wr, const float* const* w_k,
                               const int32_t* pred, const int32_t* spk, int B, int T, float* const* H1, int ldo,
                               float* const* GI, int ldgi, float* const* Mseq, float* const* GH, float* const* R,
                               float* const* ks, float* const* alpha, const int* cfg, int32_t* state, int32_t* health,
                               void* scratch, void* stream) {
    ERC_REQUIRE(H0 && Wh && bh && W_hh_c && b_hh_c && W_ih_p && b_ih_p && Wr && w_k && pred && spk && H1 && GI && Mseq && GH &&
                    R && ks && alpha && state && scratch,
                "dag_rec_fwd: null pointer");
    ERC_REQUIRE(B > 0 && T > 0 && T < 1023 && n_layers > 0 && ldh0 >= HID && ldo >= HID && ldgi > 6 * HID,
                "dag_rec_fwd: bad sizes B=%d T=%d layers=%d", B, T, n_layers);
    ERC_REQUIRE(cfg_ok(cfg) && ((uintptr_t)scratch & 7) == 0, "dag_rec_fwd: bad configuration (use erc_dag_rec_config)");
    const int dg = cfg[1], gpl = cfg[2], lpl = cfg[3];
    const int lds = lds_fwd(cfg[0], dg, T);
    ERC_REQUIRE(lds <= 160 * 1024, "dag_rec_fwd: T=%d with %d dialogues per group needs %d bytes of LDS", T, dg, lds);
    for (int l = 0; l < n_layers; ++l)
        ERC_REQUIRE(Wh[l] && bh[l] && W_hh_c[l] && b_hh_c[l] && W_ih_p[l] && b_ih_p[l] && Wr[l] && w_k[l] && H1[l] && GI[l] &&
                        Mseq[l] && GH[l] && R[l] && ks[l] && alpha[l],
                    "dag_rec_fwd: null pointer in the tables of layer %d", l);
    const int groups = erc_cdiv(B, dg);
    u64* xm = reinterpret_cast<u64*>(scratch);
    u64* xh = xm + (int64_t)groups * ML * XG;
    hipStream_t st = (hipStream_t)stream;
    for (int l0 = 0; l0 < n_layers; l0 += lpl) {
        const int nl = n_layers - l0 < lpl ? n_layers - l0 : lpl;
        for (int g0 = 0; g0 < groups; g0 += gpl) {
            const int ng = groups - g0 < gpl ? groups - g0 : gpl;
            RecFwd p;
            p.H0 = l0 == 0 ? H0 : H1[l0 - 1];           // a later chunk reads the previous chunk's output as a plain matrix
            p.ldh0 = l0 == 0 ? ldh0 : ldo;
            for (int l = 0; l < nl; ++l)
                p.ly[l] = FwdLayer{Wh[l0 + l], bh[l0 + l], W_hh_c[l0 + l], b_hh_c[l0 + l], W_ih_p[l0 + l], b_ih_p[l0 + l],
                                   Wr[l0 + l], w_k[l0 + l], H1[l0 + l], GI[l0 + l], Mseq[l0 + l], GH[l0 + l], R[l0 + l],
                                   ks[l0 + l], alpha[l0 + l]};
            for (int l = nl; l < ML; ++l) p.ly[l] = p.ly[0];
            p.ldo = ldo, p.ldgi = ldgi, p.pred = pred, p.spk = spk;
            p.B = B, p.T = T, p.DG = dg, p.g0 = g0, p.nl = nl;
            p.xm = xm, p.xh = xh, p.epoch = state + 1, p.err = health ? health : state, p.stamps = g_stamps;
            REC_DISPATCH(dag_rec_fwd_kernel, p, lds, ng * nl)
            ERC_LAUNCH_CHECK("dag_rec_fwd");
        }
    }
    return ERC_OK;
}

extern "C" int erc_dag_rec_bwd(int n_layers, const float* const* Hl, int ldh, const float* const* GI, int ldgi,
                               const float* const* GH, const float* const* Mseq, const float* const* R,
                               const float* const* alpha, const float* const* Wh, const float* const* W_hh_c,
                               const float* const* W_ih_p, const float* const* Wr, const float* const* w_k,
                               const int32_t* pred, const int32_t* spk, int B, int T, float* dHall, int ldd,
                               float* const* DGI, int lddgi, float* const* DGH, float* const* dM, float* const* dks,
                               const int* cfg, int32_t* state, int32_t* health, void* scratch, void* stream) {
    ERC_REQUIRE(Hl && GI && GH && Mseq && R && alpha && Wh && W_hh_c && W_ih_p && Wr && w_k && pred && spk && dHall && DGI && DGH &&
                    dM && dks && state && scratch,
                "dag_rec_bwd: null pointer");
    ERC_REQUIRE(B > 0 && T > 0 && T < 1021 && n_layers > 0 && ldh >= HID && ldgi > 6 * HID && lddgi > 6 * HID &&
                    ldd >= HID * (n_layers + 1),
                "dag_rec_bwd: bad sizes B=%d T=%d layers=%d", B, T, n_layers);
    ERC_REQUIRE(cfg_ok(cfg) && ((uintptr_t)scratch & 7) == 0, "dag_rec_bwd: bad configuration (use erc_dag_rec_config)");
    const int dg = cfg[1], gpl = cfg[2], lpl = cfg[3];
    const int lds = lds_bwd(cfg[0], dg, T);
    ERC_REQUIRE(lds <= 160 * 1024, "dag_rec_bwd: T=%d with %d dialogues per group needs %d bytes of LDS", T, dg, lds);
    for (int l = 0; l < n_layers; ++l)
        ERC_REQUIRE(Hl[l] && GI[l] && GH[l] && Mseq[l] && R[l] && alpha[l] && Wh[l] && W_hh_c[l] && W_ih_p[l] && Wr[l] && w_k[l] &&
                        DGI[l] && DGH[l] && dM[l] && dks[l],
                    "dag_rec_bwd: null pointer in the tables of layer %d", l);
    const int groups = erc_cdiv(B, dg);
    const int64_t P = HID / cfg[0];
    u64* xm = reinterpret_cast<u64*>(scratch);
    u64* xu = xm + (int64_t)groups * ML * XG;
    u64* xd = xu + (int64_t)groups * ML * T * XG;
    hipStream_t st = (hipStream_t)stream;
    for (int hi = n_layers; hi > 0; hi -= lpl) {          // chunks of layers, top chunk first
        const int l0 = hi - lpl > 0 ? hi - lpl : 0, nl = hi - l0;
        for (int g0 = 0; g0 < groups; g0 += gpl) {
            const int ng = groups - g0 < gpl ? groups - g0 : gpl;
            RecBwd p;
            for (int l = 0; l < nl; ++l)
                p.ly[l] = BwdLayer{Hl[l0 + l], GI[l0 + l], GH[l0 + l], Mseq[l0 + l], R[l0 + l], alpha[l0 + l], Wh[l0 + l],
                                   W_hh_c[l0 + l], W_ih_p[l0 + l], Wr[l0 + l], w_k[l0 + l], dHall + (int64_t)HID * (l0 + l + 1),
                                   DGI[l0 + l], DGH[l0 + l], dM[l0 + l], dks[l0 + l]};
            for (int l = nl; l < ML; ++l) p.ly[l] = p.ly[0];
            p.ldh = ldh, p.ldgi = ldgi, p.ldd = ldd, p.lddgi = lddgi;
            p.dLow = dHall + (int64_t)HID * l0, p.relu_low = l0 == 0;
            p.pred = pred, p.spk = spk, p.B = B, p.T = T, p.DG = dg, p.g0 = g0, p.nl = nl;
            p.xd = xd, p.xm = xm, p.xu = xu, p.epoch = state + 1, p.err = health ? health : state, p.stamps = g_stamps;
            REC_DISPATCH(dag_rec_bwd_kernel, p, lds, ng * nl)
            ERC_LAUNCH_CHECK("dag_rec_bwd");
        }
    }
    return ERC_OK;
}

extern "C" int erc_dag_attn_sums(const float* alpha, const float* H1, int ldo, const int32_t* pred, const int32_t* spk, int B,
                                 int T, float* A, void* stream) {
    ERC_REQUIRE(alpha && H1 && pred && spk && A && B > 0 && T > 0 && ldo >= HID, "dag_attn_sums: bad arguments");
    hipLaunchKernelGGL(dag_attn_sums_kernel, dim3(B * T), dim3(64), 0, (hipStream_t)stream, alpha, H1, ldo, pred, spk, T, A);
    ERC_LAUNCH_CHECK("dag_attn_sums");
    return ERC_OK;
}
