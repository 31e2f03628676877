// K6 (second generation): the DAG-ERC directed-acyclic recurrence (track_mm/dagerc.py:167-189,
// track_mm/dagerc_models.py:326-365) weight-stationary, with the batch of dialogues as the MFMA M dimension.
//
// The first-generation scan (dag_scan.hip) gave every dialogue its own cluster of workgroups and re-streamed the
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

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ u64 ld64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_tag(u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_err(const int* e) { return __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void set_err(int* e) { __hip_atomic_store(e, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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
struct RecBwd {
    const float* Hl; int ldh;
    const float* GI; int ldgi;
    const float *GH, *Mseq, *R, *alpha;
    const float *W_hh_c, *W_ih_p, *Wr, *w_k;
    const int32_t *pred, *spk;
    const float* dH1; int ldd;          // complete gradient wrt the layer outputs
    float* dHl; int lddl;               // += the direct gradient wrt H_l (z_p * g)
    float* DGI; int lddgi;              // [B*T, >= 1801] written: hoisted-side gate gradients | d(query score)
    float* DGH;                         // [B*T, 1800] written: sequential-side gate gradients
    float *dR, *dks;                    // [B*T,600], [B*T] written
    int B, T, DG, g0;
    u64 *xd, *xm;                       // [groups][P consumers][P producers][ndlg][EPC] ; [groups][300][16]
    int *epoch, *err;
    u64* stamps;
};

template <int EPC>
__global__ __launch_bounds__(NTH) void dag_rec_bwd_kernel(RecBwd p) {
    constexpr int NSL = HID / EPC;
    constexpr int NKG = (6 * EPC + 3) / 4;              // k-steps of the transposed gate product (this slice's rows)
    constexpr int VP = 4 * NKG + 1;
    constexpr int MAXP = (MAXDG * EPC + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = p.g0 + blockIdx.x / NSL, c = blockIdx.x % NSL;
    const int DG = p.DG, T = p.T;
    const int b0 = grp * DG, ndlg = min(DG, p.B - b0);
    float* vin = smem;                                  // [16][VP]   sequential-side gate gradients of this slice
    float* red = vin + 16 * VP;                         // [NTH]
    float* part_y = red + NTH;                          // [NMW][16][PST]
    float* dmfull = part_y + NMW * 16 * PST;            // [16][DMP]  the gathered dM_i
    float* gacc = dmfull + 16 * DMP;                    // [T][DG][EPC]    sum_i alpha_ij (Wr_sel^T dM_i)[E_c]
    float* dracc = gacc + T * DG * EPC;                 // [T][DG][2 EPC]  dR
    float* dks_s = dracc + T * DG * 2 * EPC;            // [DG][T]
    float* dal = dks_s + DG * T;                        // [DG][T]         dalpha of the current step's window
    int* s_spk = reinterpret_cast<int*>(dal + DG * T);
    int* s_pred = s_spk + DG * T;
    for (int x = tid; x < ndlg * T; x += NTH) {
        s_spk[x] = p.spk[(int64_t)b0 * T + x];
        s_pred[x] = p.pred[(int64_t)b0 * T + x];
    }
    for (int x = tid; x < 16 * VP; x += NTH) vin[x] = 0.f;
    for (int x = tid; x < T * DG * (3 * EPC + 1); x += NTH) gacc[x] = 0.f;      // gacc | dracc | dks_s are contiguous
    const unsigned ep = (unsigned)p.epoch[grp] + 1u;
    u64* const xm = p.xm + (int64_t)grp * XG;
    const int IT = ndlg * EPC;                            // records per producer block
    u64* const xd = p.xd + (int64_t)grp * NSL * NSL * DG * EPC;

    // ---- stationary weights
    float wgT[TPW][NKG], wrY[NQ];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int t = wave + NMW * u, k = 16 * t + (lane & 15), kc = min(k, HID - 1);
        const float kv = (wave < NMW && t < NT19 && k < HID) ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < NKG; ++s) {
            const int n = 4 * s + (lane >> 4), nc = min(n, 6 * EPC - 1);
            const float v = gate_row(p.W_hh_c, p.W_ih_p, nc / EPC, c * EPC + nc % EPC)[kc];
            wgT[u][s] = v * (n < 6 * EPC ? kv : 0.f);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int s = wave + NMW * q, ec = min(4 * s + (lane >> 4), HID - 1);
        const int j = lane & 15, jc = min(j, 2 * EPC - 1);
        const float v = p.Wr[(int64_t)((jc / EPC) * HID + ec) * HID + c * EPC + jc % EPC];
        wrY[q] = v * ((wave < NMW && s < KS && j < 2 * EPC) ? 1.f : 0.f);
    }
    bool iv[MAXP];
    int im[MAXP], iel[MAXP];
    float wk_e[MAXP], dmdir[MAXP];
#pragma unroll
    for (int r = 0; r < MAXP; ++r) {
        const int it = lane + 64 * r;
        iv[r] = wave == EWW && it < IT;
        const int itc = iv[r] ? it : 0;
        iel[r] = itc / ndlg, im[r] = itc % ndlg;
        wk_e[r] = p.w_k[c * EPC + iel[r]];
        dmdir[r] = 0.f;
    }
    __syncthreads();

    for (int i = T - 1; i >= 0; --i) {
        const unsigned tag = ep * 1024u + (unsigned)i + 1u;
        REC_STAMP(0);
        // ---- E1: total gradient wrt h_i for this slice, GRU cells backward (elementwise)
        if (wave == EWW) {
#pragma unroll
            for (int r = 0; r < MAXP; ++r) {
                if (!iv[r]) continue;
                const int m = im[r], el = iel[r], e = c * EPC + el;
                const int64_t row = (int64_t)(b0 + m) * T + i;
                const float* gi = p.GI + row * p.ldgi;
                const float* gh = p.GH + row * 6 * HID;
                float giv[6], ghv[6];
#pragma unroll
                for (int g = 0; g < 6; ++g) giv[g] = gi[g * HID + e], ghv[g] = gh[g * HID + e];
                const float mi = p.Mseq[row * HID + e], xi = p.Hl[row * p.ldh + e];
                const float g = p.dH1[row * p.ldd + e] + gacc[(i * DG + m) * EPC + el] + wk_e[r] * dks_s[m * T + i];
                float* dgi = p.DGI + row * p.lddgi;
                float* dgh = p.DGH + row * 6 * HID;
                float* vrow = vin + m * VP;
                {   // cell C: x = H_l[i] (hoisted side), h = M_i (sequential side)
                    const float rr = sigm(giv[0] + ghv[0]), zz = sigm(giv[1] + ghv[1]);
                    const float nn = tanhf(giv[2] + rr * ghv[2]);
                    const float dn = g * (1.f - zz) * (1.f - nn * nn);
                    const float dz = g * (mi - nn) * zz * (1.f - zz);
                    const float dr = dn * ghv[2] * rr * (1.f - rr);
                    dgi[e] = dr, dgi[HID + e] = dz, dgi[2 * HID + e] = dn;
                    dgh[e] = dr, dgh[HID + e] = dz, dgh[2 * HID + e] = dn * rr;
                    vrow[el] = dr, vrow[EPC + el] = dz, vrow[2 * EPC + el] = dn * rr;
                    dmdir[r] = g * zz;                                    // direct path into M_i
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
                    p.dHl[row * p.lddl + e] += g * zz;                    // direct path into H_l[i]
                }
                if (i == 0 && c == 0 && el == 0) dgi[6 * HID] = 0.f;      // step 0 has no attention: d(query score) = 0
            }
        }
        if (i == 0) break;                                                // M_0 = 0 has no producers
        REC_STAMP(1);
        __syncthreads();
        // ---- M1: partial dM_i = Wg[rows of E_c]^T dgates, one partial FULL vector per dialogue -> reduce-scatter
        if (wave < NMW) {
            float a[NKG];
#pragma unroll
            for (int s = 0; s < NKG; ++s) a[s] = vin[(lane & 15) * VP + 4 * s + (lane >> 4)];
            f32x4 acc[TPW];
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NKG; ++s) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wgT[u][s], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const int k = 16 * (wave + NMW * u) + (lane & 15);
                const int cc = k / EPC, el2 = k % EPC;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = 4 * (lane >> 4) + r;
                    if (wave + NMW * u < NT19 && k < HID && mm < ndlg)
                        st_tag(xd + ((int64_t)(cc * NSL + c) * ndlg + mm) * EPC + el2, acc[u][r], tag);
                }
            }
        }
        REC_STAMP(2);
        {   // every thread sums its share of the P partial blocks addressed to this slice, in producer order
            const int S = (NTH / IT) * IT, total = NSL * IT;
            const u64* blk = xd + (int64_t)c * NSL * IT;
            float sum = 0.f;
            if (tid < S) {
                int spins = 0;
                if (tid < total) {   // sentinel: wait for this thread's first record before requesting the others (see poll_operand)
                    u64 v0 = ld64(blk + tid);
                    while ((unsigned)(v0 >> 32) != tag) {
                        if (++spins > SPIN_LIMIT) {
                            set_err(p.err);
                            break;
                        }
                        if ((spins & 255) == 0 && ld_err(p.err)) break;
                        __builtin_amdgcn_s_sleep(POLL_SLEEP);
                        v0 = ld64(blk + tid);
                    }
                }
                for (int x0 = tid; x0 < total; x0 += 4 * S) {
                    u64 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int x = x0 + u * S;
                        v[u] = (u64)tag << 32;
                        if (x < total) v[u] = ld64(blk + x);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) sum += settle(blk + min(x0 + u * S, total - 1), v[u], tag, spins, p.err);
                }
            }
            red[tid] = sum;
        }
        REC_STAMP(3);
        __syncthreads();
        // ---- E2 / M2: dM_i of this slice -> all-gather; Y_i = Wr[:, E_c]^T dM_i
        if (wave == EWW) {
            const int S = (NTH / IT) * IT;
#pragma unroll
            for (int r = 0; r < MAXP; ++r) {
                if (!iv[r]) continue;
                const int m = im[r], el = iel[r], e = c * EPC + el;
                float dm = dmdir[r];
                for (int x = m * EPC + el; x < S; x += IT) dm += red[x];
                st_tag(xm + e * XROW + m, dm, tag);
                const int pr = s_pred[m * T + i], lo = pr > 0 ? pr : 0, si = s_spk[m * T + i];
                const float* arow = p.alpha + ((int64_t)(b0 + m) * T + i) * T;
                for (int j = lo; j < i; ++j)                               // dV_j = alpha_ij dM_i into the slot that was read
                    dracc[(j * DG + m) * 2 * EPC + (s_spk[m * T + j] == si ? 0 : EPC) + el] += arow[j] * dm;
            }
        } else {
            float a[NQ];
            poll_operand<NMW, NQ>(xm, wave, lane, ndlg, tag, a, p.err);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int s = wave + NMW * q;
                if (s < KS) dmfull[(lane & 15) * DMP + 4 * s + (lane >> 4)] = a[q];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wrY[q], acc, 0, 0, 0);
            }
            store_tile(part_y + wave * 16 * PST, acc, lane);
        }
        REC_STAMP(4);
        __syncthreads();
        REC_STAMP(5);
        // ---- dalpha_ij = dM_i . V_j over the window (V_j = the relation row of j that step i read)
        if (wave < NMW) {
            int nmax = 0;
            for (int m = 0; m < ndlg; ++m) {
                const int pr = s_pred[m * T + i];
                nmax = max(nmax, i - (pr > 0 ? pr : 0));
            }
            for (int d = wave; d < ndlg * nmax; d += NMW) {
                const int m = d % ndlg, jj = d / ndlg;
                const int pr = s_pred[m * T + i], lo = pr > 0 ? pr : 0;
                if (jj >= i - lo) continue;
                const int j = lo + jj;
                const float* v = p.R + ((int64_t)(b0 + m) * T + j) * 2 * HID + (s_spk[m * T + j] == s_spk[m * T + i] ? 0 : HID);
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const int k = lane + 64 * r, kc = min(k, HID - 1);
                    s += v[kc] * dmfull[m * DMP + kc] * (k < HID ? 1.f : 0.f);
                }
                s = wave_sum(s);
                if (lane == 0) dal[m * T + jj] = s;
            }
        }
        REC_STAMP(6);
        __syncthreads();
        // ---- E3: softmax backward, accumulations for the earlier steps
        if (wave == EWW) {
#pragma unroll
            for (int r = 0; r < MAXP; ++r) {
                if (!iv[r]) continue;
                const int m = im[r], el = iel[r];
                float y0 = 0.f, y1 = 0.f;
#pragma unroll
                for (int w = 0; w < NMW; ++w) {
                    const float* py = part_y + (w * 16 + m) * PST;
                    y0 += py[el], y1 += py[EPC + el];
                }
                const int pr = s_pred[m * T + i], lo = pr > 0 ? pr : 0, si = s_spk[m * T + i], n = i - lo;
                const float* arow = p.alpha + ((int64_t)(b0 + m) * T + i) * T + lo;
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
                if (el == 0 && c == 0) p.DGI[((int64_t)(b0 + m) * T + i) * p.lddgi + 6 * HID] = dq;
            }
        }
        REC_STAMP(7);
    }
    __syncthreads();
    for (int x = tid; x < T * ndlg * 2 * EPC; x += NTH) {
        const int el2 = x % (2 * EPC), m = (x / (2 * EPC)) % ndlg, j = x / (2 * EPC * ndlg);
        p.dR[((int64_t)(b0 + m) * T + j) * 2 * HID + (el2 / EPC) * HID + c * EPC + el2 % EPC] = dracc[(j * DG + m) * 2 * EPC + el2];
    }
    if (c == 0)
        for (int x = tid; x < T * ndlg; x += NTH) p.dks[(int64_t)(b0 + x / T) * T + x % T] = dks_s[x];
    if (c == 0 && tid == 0) p.epoch[grp] = (int)ep;
}

// ----------------------------------------------------------------------------------------------- host side
int lds_fwd(int epc, int dg, int T) {
    const int ntg = (6 * epc + 15) / 16, nth2 = (6 * epc + 1 + 15) / 16;
    return 4 * (FMW * (ntg + nth2 + 1) * 16 * PST + T * dg * 2 * epc + T * dg + 2 * dg * T);
}
int lds_bwd(int epc, int dg, int T) {
    const int nkg = (6 * epc + 3) / 4;
    return 4 * (16 * (4 * nkg + 1) + NTH + NMW * 16 * PST + 16 * DMP + T * dg * (3 * epc + 1) + dg * T + 2 * dg * T);
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

// workgroups of a kernel that the device holds at once (one per CU is what the exchange latency is tuned for; never more
// than the occupancy query admits); also raises the kernel's dynamic-LDS limit to what the launch will ask for
template <typename K>
int capacity(K kernel, int lds) {
    const int cus = device_cus();
    if (cus < 0) return -1;
    if (lds > 160 * 1024) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
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
            const int lmax = dir ? 1 : (n_layers < ML ? n_layers : ML);      // only the forward pipelines the layers
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
// [groups][4][T][300][16]; backward: all-gather records of dM [groups][300][16], reduce-scatter records
// [groups][P][P][dg][epc] (8 bytes each)
extern "C" int64_t erc_dag_rec_scratch_bytes(int dir, int B, int T, const int* cfg) {
    if (B <= 0 || T <= 0 || !cfg_ok(cfg)) return -1;
    const int64_t groups = erc_cdiv(B, cfg[1]), P = HID / cfg[0];
    return dir ? 8 * (groups * XG + groups * P * P * cfg[1] * cfg[0]) : 8 * (groups * ML * XG * (1 + (int64_t)T));
}

#define REC_DISPATCH(KERNEL, ARGS, LDS, WGS)                                                                    \
    switch (cfg[0]) {                                                                                           \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3((WGS) * (HID / 2)), dim3(NTH), LDS, st, ARGS); break;        \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3((WGS) * (HID / 4)), dim3(NTH), LDS, st, ARGS); break;        \
        case 5: hipLaunchKernelGGL(KERNEL<5>, dim3((WGS) * (HID / 5)), dim3(NTH), LDS, st, ARGS); break;        \
    }

extern "C" int erc_dag_rec_fwd(const float* H0, int ldh0, int n_layers, const float* const* Wh, const float* const* bh,
                               const float* const* W_hh_c, const float* const* b_hh_c, const float* const* W_ih_p,
                               const float* const* b_ih_p, const float* const* Wr, const float* const* w_k,
                               const int32_t* pred, const int32_t* spk, int B, int T, float* const* H1, int ldo,
                               float* const* GI, int ldgi, float* const* Mseq, float* const* GH, float* const* R,
                               float* const* ks, float* const* alpha, const int* cfg, int32_t* state, void* scratch,
                               void* stream) {
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
            p.xm = xm, p.xh = xh, p.epoch = state + 1, p.err = state, p.stamps = g_stamps;
            REC_DISPATCH(dag_rec_fwd_kernel, p, lds, ng * nl)
            ERC_LAUNCH_CHECK("dag_rec_fwd");
        }
    }
    return ERC_OK;
}

extern "C" int erc_dag_rec_bwd(const float* Hl, int ldh, const float* GI, int ldgi, const float* GH, const float* Mseq,
                               const float* R, const float* alpha, const float* W_hh_c, const float* W_ih_p, const float* Wr,
                               const float* w_k, const int32_t* pred, const int32_t* spk, int B, int T, const float* dH1,
                               int ldd, float* dHl, int lddl, float* DGI, int lddgi, float* DGH, float* dR, float* dks,
                               const int* cfg, int32_t* state, void* scratch, void* stream) {
    ERC_REQUIRE(Hl && GI && GH && Mseq && R && alpha && W_hh_c && W_ih_p && Wr && w_k && pred && spk && dH1 && dHl && DGI &&
                    DGH && dR && dks && state && scratch,
                "dag_rec_bwd: null pointer");
    ERC_REQUIRE(B > 0 && T > 0 && T < 1023 && ldh >= HID && ldgi > 6 * HID && lddgi > 6 * HID, "dag_rec_bwd: bad sizes B=%d T=%d", B, T);
    ERC_REQUIRE(cfg_ok(cfg) && ((uintptr_t)scratch & 7) == 0, "dag_rec_bwd: bad configuration (use erc_dag_rec_config)");
    const int dg = cfg[1], gpl = cfg[2];
    const int lds = lds_bwd(cfg[0], dg, T);
    ERC_REQUIRE(lds <= 160 * 1024, "dag_rec_bwd: T=%d with %d dialogues per group needs %d bytes of LDS", T, dg, lds);
    const int groups = erc_cdiv(B, dg);
    u64* xm = reinterpret_cast<u64*>(scratch);
    u64* xd = xm + (int64_t)groups * XG;
    hipStream_t st = (hipStream_t)stream;
    for (int g0 = 0; g0 < groups; g0 += gpl) {
        const int ng = groups - g0 < gpl ? groups - g0 : gpl;
        RecBwd p{Hl, ldh, GI, ldgi, GH, Mseq, R, alpha, W_hh_c, W_ih_p, Wr, w_k, pred, spk, dH1, ldd, dHl, lddl, DGI, lddgi,
                 DGH, dR, dks, B, T, dg, g0, xd, xm, state + 1, state, g_stamps};
        REC_DISPATCH(dag_rec_bwd_kernel, p, lds, ng)
        ERC_LAUNCH_CHECK("dag_rec_bwd");
    }
    return ERC_OK;
}
