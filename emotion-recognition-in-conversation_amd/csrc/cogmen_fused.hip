// COGMEN graph part as row-tile kernels (bf16 compute mode): everything between the input projection and the BatchNorm
// of track_mm/cogmen.py:61-74 (GNN.forward: RGCNConv -> TransformerConv) in ONE launch, and its backward in one more.
//
// The dialogue graph is a window graph (cogmen.py:153-154, wp = wf = 5): node i only sees rows [i-5, i+5] of its own
// dialogue, so a tile of 16 consecutive nodes needs a HALO, not a global exchange.  A workgroup computes
//   forward  : M = [mean_r H0 | H0] for rows [r0-5, r0+21)  -> H1 = M Wcat + b (those 26 rows) -> QKVS = H1 Wqkvs^T + b
//              (26 rows) -> segmented-softmax attention + skip for its 16 rows -> H2, plus the BatchNorm column sums
//              (partials per tile, finalised by the last workgroup to arrive)
//   backward : dH2 = BatchNorm backward for rows [r0-10, r0+26) -> attention backward as fp32 band products on the matrix
//              cores (dA = G V^T on the +-5 band, d(score), dq = DS K, dk = DS^T Q, dv = AL^T G) -> dH1 = dQKVS Wqkvs
//              (26 rows) -> dP = A^T dH1 (16 rows, the transposed relation means) -> dH0 = dP [W_r^T] (16 rows)
// with the intermediate tiles in LDS.  The halo rows are recomputed by the neighbouring tile (1.6x .. 2.3x of the small
// products) -- that is what buys 5 + 5 launches less and no HBM / L2 round trip for M, H1 (fp32), dM.
//
// The dense products run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: A fragments from the LDS tile
// (ds_read_b128, rows 16-byte aligned), B fragments straight from bf16 shadow copies of the weights that the optimizer
// launch keeps in sync (erc_adam_step_tab), stored in the fragment order of the instruction's B operand (ErcShadowTab
// mode 1: the 512 elements of a (16-column tile, 32-deep K block) are contiguous, so a wavefront's fragment load is one
// 1 KB run) -- logical operands [n][k]:
//   WcatT [112][928]  WcatT[o][r*100+c] = W_r[c][o]  (r = 8: root)       H1 = M Wcat
//   Wq    [400][128]  the [q;k;v;skip] Linear weights, K padded           QKVS = H1 Wq^T
//   WqT   [112][416]  WqT[c][n] = Wq[n][c]                                dH1 = dQKVS Wq
//   Wb    [112][960]  Wb[c][r*104+o] = W_r[c][o]                          dH0 = dP [W_r^T]
// Everything else (means, softmax, BatchNorm) is fp32.  The graph is read through the CSR that erc_window_graph_build
// wrote; the kernels only rely on |source - target| <= 5 (checked by the host wrapper through wp / wf).
#include "erc_common.h"
#include "split_dev.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CG_F = 100;                    // channels (the reference hard-codes 100: cogmen.py:116-122)
constexpr int CG_R = 8;                      // relations (GNN(n_speakers = 2): cogmen.py:62-64)
constexpr int CG_HL = 5;                     // halo = max(wp, wf)
constexpr int CG_TR = 16;                    // rows a tile owns
constexpr int CG_MID = CG_TR + 2 * CG_HL;    // 26
constexpr int CG_OUT = CG_TR + 4 * CG_HL;    // 36
constexpr int CG_FAR = CG_TR + 6 * CG_HL;    // 46
constexpr int CG_KM = (CG_R + 1) * CG_F;     // 900
constexpr int CG_KMP = 928;                  // K of the H1 product, padded to 29 blocks of 32
constexpr int CG_SM = 936;                   // LDS pitch of the M tile (bf16 elements; 1872 B = 117 x 16)
constexpr int CG_SH1 = 136;                  // LDS pitch of the H1 tile (bf16; K padded to 128)
constexpr int CG_SQ = 404;                   // LDS pitch of the QKVS tile (floats)
constexpr int CG_CH = 12;                    // neighbour rows handled per batch (window graph: <= 11 in-edges)
constexpr int CG_NT = 7;                     // column tiles of 16 covering F = 100

constexpr int CG_NW = 16;                    // wavefronts per workgroup (1024 threads: <= 128 VGPRs)
constexpr int CG_NTH = 64 * CG_NW;

// LDS map of the forward kernel (bytes)
constexpr int FW_SM_OFF = 0;                                   // M tile, 27 rows (26 + one zero row); later the QKVS tile (26 x 404 floats)
constexpr int FW_SM_BYTES = 27 * CG_SM * 2;                    // 50544  (26 * 404 * 4 = 42016 fits)
constexpr int FW_SH1_OFF = FW_SM_OFF + FW_SM_BYTES;
constexpr int FW_SH1_BYTES = 32 * CG_SH1 * 2;                  // 8704
constexpr int FW_SH0_OFF = FW_SH1_OFF + FW_SH1_BYTES;          // H0 rows [r0-10, r0+26) fp32; then the K-split partials of
constexpr int FW_SH0_BYTES = CG_OUT * CG_F * 4;                //   the H1 product (7 x 64 x 8 floats = 14336); then the H2 tile
constexpr int FW_ECAP = 288;                                   // in-edges of the 26 mid rows (<= 26 x 11)
constexpr int FW_SE_OFF = FW_SH0_OFF + FW_SH0_BYTES;           // sSrc[288], sTyp[288], sIp[32]
constexpr int FW_SE_BYTES = (2 * FW_ECAP + 32) * 4;
constexpr int FW_RED_OFF = FW_SE_OFF + FW_SE_BYTES;            // BatchNorm finalisation scratch: 4 x 256 doubles
constexpr int FW_RED_BYTES = 4 * 256 * 8;
constexpr int FW_PFX_OFF = FW_RED_OFF + FW_RED_BYTES + 16;     // two-speaker path: exclusive per-speaker prefix sums of the H0 rows
constexpr int FW_PFX_BYTES = 2 * (CG_OUT + 1) * CG_F * 8;      //   [2][37][100] fp64 = 59200
constexpr int FW_SPK_OFF = FW_PFX_OFF + FW_PFX_BYTES;          // speakers of the 36 outer rows
constexpr int FW_LDS = FW_SPK_OFF + 64 * 4;
static_assert(FW_PFX_OFF % 16 == 0 && FW_LDS <= 160 * 1024, "forward LDS map");
static_assert(7 * 64 * 8 * 4 <= FW_SH0_BYTES, "K-split partials must fit the H0 area");

// ---- SPLIT COMPUTE MODES (NT = 2, 3 bf16 terms per fp32 operand value; csrc/split_dev.h): LDS map of the forward kernel.
// Two-speaker graphs only (the reference's GNN(n_speakers = 2), cogmen.py:62-64): a target with speaker a has in-edges of
// relations 2 a + {0, 1, 4, 5} only, so its row of M holds FIVE non-empty blocks (four relation means + self).  The M tile is
// stored COMPACT, [26 rows][5 x 100] per term plane, and the H1 product still runs over the K = 928 of the weight shadow: a
// lookup table maps every 4-element group of K to its compact column and the speaker it belongs to, rows of the other
// speaker (and the K padding) read a zero slot.  NT planes of the full [27][936] tile would not fit next to the fp64 prefix sums.
constexpr int FX_CM = 520;                                     // pitch of the compact M tile (bf16 elements; 1040 B rows)
constexpr int FX_MPLANE = CG_MID * FX_CM * 2;                  // 27040 bytes per term plane
constexpr int FX_ZERO = 2 * 512;                               // plane-relative byte offset of 8 zero bytes (row 0, columns [512, 516): never written by the aggregation)
// weight fragments of the H1 product requested ahead (K blocks per wavefront, of its 15).  Measured: 2 ahead 21.3 / 27.0 us per launch
// (2 / 3 terms), 7 / 5 ahead -- everything the registers hold across the aggregation -- 22.3 / 27.5: a tile's NT x 0.3 MB of weight planes
// come through the per-CU L2 path (~70 GB/s) whenever they are requested, and a burst in front of a stage delays that stage's own loads
// (DESIGN.md finding 20)
template <int NT> constexpr int FX_PF = 3;
template <int NT>
struct FxMap {
    static constexpr int SM_OFF = 0;                           // M planes; the fp64 segment totals of the prefix scan before them; later the QKVS tile
    static constexpr int SM_BYTES = NT * FX_MPLANE > CG_MID * CG_SQ * 4 ? NT * FX_MPLANE : CG_MID * CG_SQ * 4;
    static constexpr int SH0_OFF = SM_OFF + SM_BYTES;          // H0 rows, then the K-split partials, then the H2 tile (as FW_SH0)
    static constexpr int SE_OFF = SH0_OFF + FW_SH0_BYTES;
    static constexpr int PFX_OFF = ((SE_OFF + FW_SE_BYTES + 15) / 16) * 16;   // prefix sums; after the aggregation: H1 planes | BatchNorm scratch
    static constexpr int SH1_OFF = PFX_OFF;
    static constexpr int RED_OFF = PFX_OFF + 3 * FW_SH1_BYTES;
    static constexpr int SPK_OFF = PFX_OFF + FW_PFX_BYTES;
    static constexpr int LUT_OFF = SPK_OFF + 256;              // 256 x uint32: compact byte offset | needed speaker << 16
    static constexpr int LDS = LUT_OFF + 1024;
    static_assert(RED_OFF + FW_RED_BYTES + 16 <= PFX_OFF + FW_PFX_BYTES && PFX_OFF % 16 == 0 && LDS <= 160 * 1024, "split forward LDS map");
    static_assert(2048 + 4 * 2 * CG_F * 8 <= FX_MPLANE && FX_ZERO + 8 <= 2048, "segment totals of the prefix scan live in the M area, clear of the zero slots");
};

__device__ __forceinline__ unsigned short f2bf(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ double ld_sc1d(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1d(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// CG_CH (<= 16) per-lane partial dot products -> every lane gets all wave totals: one 16-value butterfly
__device__ __forceinline__ void wave_sums_ch(const float (&part)[16], float (&tot)[CG_CH], int lane) {
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
    e += __shfl_xor(e, 1, 64);
#pragma unroll
    for (int u = 0; u < CG_CH; ++u) tot[u] = __shfl(e, ((u >> 3) & 1) * 32 + ((u >> 2) & 1) * 16 + ((u >> 1) & 1) * 8 + (u & 1) * 4, 64);
}

// diagnostic phase stamps (tools/cogmen_stamps.py): the 100 MHz real-time counter, written by thread 0 of one workgroup
#define CG_STAMP(slot)                                                                                            \
    do {                                                                                                          \
        if (p.stamps && (int)blockIdx.x == p.stamp_block && threadIdx.x == 0) p.stamps[slot] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// one of 8 accumulator pairs selected by a WAVE-UNIFORM relation id: a scalar branch, two adds executed
#define CG_ACC8(ty, s0, s1, va, vb)                          \
    do {                                                     \
        switch (ty) {                                        \
            case 0: s0[0] += va, s1[0] += vb; break;         \
            case 1: s0[1] += va, s1[1] += vb; break;         \
            case 2: s0[2] += va, s1[2] += vb; break;         \
            case 3: s0[3] += va, s1[3] += vb; break;         \
            case 4: s0[4] += va, s1[4] += vb; break;         \
            case 5: s0[5] += va, s1[5] += vb; break;         \
            case 6: s0[6] += va, s1[6] += vb; break;         \
            case 7: s0[7] += va, s1[7] += vb; break;         \
            default: break;                                  \
        }                                                    \
    } while (0)

struct CgFwdP {
    const float* H0;               // [N, ldh0] fp32: the projected utterance features
    const int32_t* in_ptr;         // CSR by target
    const int32_t* in_src;
    const int32_t* in_typ;
    const unsigned short* WcatT;   // bf16, fragment order (7 column tiles x 29 K blocks x 512)
    const float* b1;               // conv1.bias [100]
    const unsigned short* Wq;      // bf16, fragment order (25 x 4 x 512)
    const float* bq;               // [400]
    unsigned short* Mb;            // out bf16 [N, ldmb]: the relation means | self (operand of the weight gradient)
    float* inv_cnt;                // out [N, 8]
    unsigned short* H1b;           // out bf16 [N, ldh1b]
    float* QKVS;                   // out [N, 400]
    float* H2;                     // out [N, ldh2]
    float* alpha;                  // out [E]
    const int32_t* node_spk;       // speaker of every node; two_spk: all in {0, 1} (n_speakers = 2) -> prefix-sum aggregation
    double* bn_part;               // [tiles][200] column sums of H2 and H2^2 (bn_fused)
    int* bn_counter;
    float* running_mean;
    float* running_var;
    float* saved;                  // out [0,F) mean, [F,2F) rstd
    float momentum, eps, scale;
    int N, ldh0, ldmb, ldh1b, ldh2, bn_fused, two_spk;
    const int32_t* n_dev;          // capacity mode: the true node count lives on the device (N is the capacity the grid is sized for)
    int32_t* health;               // training step: the health word (erc_health_roll's contract) and its event counter, or null --
    int32_t* events;               //   this launch is the first of the step that may precede a reader of the word
    uint64_t* stamps;
    int stamp_block;
    // split compute modes: WcatT / Wq are NT term planes, *_plane elements apart; the weight-gradient operands are written as fp32
    int64_t catT_plane, q_plane;
    float* Mf;                     // out fp32 [N, ldmf >= 900]
    float* H1f;                    // out fp32 [N, ldh1f >= 100]
    int ldmf, ldh1f;
};

template <int NT>
__global__ __launch_bounds__(CG_NTH) void cogmen_fwd_tile_kernel(const CgFwdP p) {
    constexpr bool X = NT > 1;
    using XM = FxMap<X ? NT : 2>;
    // erc_health_roll folded into this launch: a word the PREVIOUS step left raised (its update was skipped) becomes one counted
    // event and is cleared, before any launch of this step reads it (the weight-gradient / optimizer launches further down)
    if (p.health && blockIdx.x == 0 && threadIdx.x == 0 && *p.health != 0) {
        p.events[0] += 1;
        *p.health = 0;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned short* const sM = reinterpret_cast<unsigned short*>(lds + (X ? XM::SM_OFF : FW_SM_OFF));
    float* const sQ = reinterpret_cast<float*>(lds + (X ? XM::SM_OFF : FW_SM_OFF));       // aliases sM (after the H1 product)
    unsigned short* const sH1 = reinterpret_cast<unsigned short*>(lds + (X ? XM::SH1_OFF : FW_SH1_OFF));
    float* const sH0 = reinterpret_cast<float*>(lds + (X ? XM::SH0_OFF : FW_SH0_OFF));
    float* const sPart = sH0;                                           // aliases sH0 (after the aggregation)
    float* const sH2 = sH0;                                             // aliases sH0 (after the H1 product)
    int* const sSrc = reinterpret_cast<int*>(lds + (X ? XM::SE_OFF : FW_SE_OFF));
    int* const sTyp = sSrc + FW_ECAP;
    int* const sIp = sTyp + FW_ECAP;
    double* const sRed = reinterpret_cast<double*>(lds + (X ? XM::RED_OFF : FW_RED_OFF));
    int* const s_last = reinterpret_cast<int*>(lds + (X ? XM::RED_OFF : FW_RED_OFF) + FW_RED_BYTES);
    double* const sPfx = reinterpret_cast<double*>(lds + (X ? XM::PFX_OFF : FW_PFX_OFF));
    int* const sSpk = reinterpret_cast<int*>(lds + (X ? XM::SPK_OFF : FW_SPK_OFF));
    uint32_t* const sLut = reinterpret_cast<uint32_t*>(lds + XM::LUT_OFF);               // (split modes)
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too: scalar branches below
    const int N = p.n_dev ? min(max(*p.n_dev, 1), p.N) : p.N;   // (uniform scalar load; rows >= N of a capacity-sized grid are masked)
    const int r0 = (int)blockIdx.x * CG_TR;
    const int mb = r0 - CG_HL, ob = r0 - 2 * CG_HL;   // first node of the mid / outer row ranges
    const int c1 = min(lane + 64, CG_F - 1);
    const bool h1 = lane + 64 < CG_F;
    CG_STAMP(0);

    // ---- stage 0: H0 rows [r0-10, r0+26) and the CSR slice of the mid rows -> LDS; zero the K padding of the M tile
    const int m_lo = min(max(mb, 0), N), m_hi = min(max(mb + CG_MID, 0), N);
    const int E_lo = p.in_ptr[m_lo], E_hi = min(p.in_ptr[m_hi], E_lo + FW_ECAP);
    // B fragments of the H1 product: wavefront (ct = w & 7, kh = w >> 3) owns column tile ct and the K blocks
    // [15 kh, 15 kh + 15) (kh = 1: 14); requested now, they arrive during the aggregation
    const int ct = w & 7, kh = w >> 3;
    const bool mma_wave = ct < CG_NT;
    constexpr int NKB = CG_KMP / 32, KH0 = 15;      // 29 blocks: 15 + 14
    const unsigned short* const brow = p.WcatT + ((int64_t)(min(ct, CG_NT - 1) * NKB + KH0 * kh) * 64 + lane) * 8;   // fragment (ct, kb) = 512 contiguous elements
    const int nkb = kh ? NKB - KH0 : KH0;       // 14 | 15
    {
        if (tid < CG_OUT * 25) {
            const int e = tid / 25, q = tid % 25;
            const int node = ob + e;
            const bool ok = node >= 0 && node < N;
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.H0 + (int64_t)min(max(node, 0), N - 1) * p.ldh0 + 4 * q);
            const float m = ok ? 1.f : 0.f;
            *reinterpret_cast<f32x4*>(sH0 + e * CG_F + 4 * q) = (f32x4){v.x * m, v.y * m, v.z * m, v.w * m};
        }
        if (tid < CG_MID + 1) sIp[tid] = p.in_ptr[min(max(mb + tid, 0), N)];
        if (tid >= 64 && tid < 64 + FW_ECAP) {
            const int i = tid - 64;
            const int ei = min(E_lo + i, max(E_hi - 1, E_lo));
            sSrc[i] = p.in_src[ei], sTyp[i] = p.in_typ[ei];
        }
        if (tid >= 960 && tid < 960 + CG_OUT) {      // speakers of the outer rows (-1: no such node)
            const int node = ob + tid - 960;
            const int sp = p.node_spk[min(max(node, 0), N - 1)];
            sSpk[tid - 960] = (node >= 0 && node < N) ? sp : -1;
        }
        if constexpr (!X) {
            // columns [900, 936) of rows 0..25: 18 dwords each; row 26 entirely: 468 dwords
            uint32_t* const sMw = reinterpret_cast<uint32_t*>(sM);
            if (tid < 26 * 18) sMw[(tid / 18) * (CG_SM / 2) + CG_KM / 2 + (tid % 18)] = 0u;
            if (tid >= 512 && tid < 512 + CG_SM / 2) sMw[26 * (CG_SM / 2) + tid - 512] = 0u;
        } else {
            // the K -> compact column table (one entry per 4 k: compact BYTE offset | needed target speaker + 1 << 16; 0: any
            // speaker (self block), 3: none (K padding)) and the zero slot of every plane
            if (tid >= 512 && tid < 768) {
                const int k4 = 4 * (tid - 512), q = k4 / CG_F, within = k4 - q * CG_F;
                uint32_t e = 3u << 16;
                if (q < CG_R) e = (uint32_t)(2 * ((((q >> 2) << 1) | (q & 1)) * CG_F + within)) | ((1u + ((q >> 1) & 1)) << 16);
                else if (q == CG_R) e = (uint32_t)(2 * (4 * CG_F + within));
                sLut[tid - 512] = e;
            }
            if (tid >= 768 && tid < 768 + 2 * NT) reinterpret_cast<uint32_t*>(lds + XM::SM_OFF + ((tid - 768) >> 1) * FX_MPLANE + FX_ZERO)[tid & 1] = 0u;
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // the tile / graph loads above are queued first
    bf16x8 bx[X ? 1 : KH0];
    sp_u32x4 bxr[X ? FX_PF<NT> : 1][X ? NT : 1];      // split modes: a ring of FX_PF K blocks x NT term planes
    if constexpr (!X) {
#pragma unroll
        for (int u = 0; u < KH0; ++u) bx[u] = *reinterpret_cast<const bf16x8*>(brow + 512 * min(u, nkb - 1));
    } else {
#pragma unroll
        for (int u = 0; u < FX_PF<NT>; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) bxr[u][t] = *reinterpret_cast<const sp_u32x4*>(brow + t * p.catT_plane + 512 * min(u, nkb - 1));
    }
    __syncthreads();
    CG_STAMP(1);

    // ---- stage A: relation means of the 26 mid rows: one wavefront per row, lane l < 50 owns channels 2l, 2l + 1.
    //      These gather stages are bound by the NUMBER of wave instructions (16 wavefronts x 4 cycles each on 4 SIMDs),
    //      not by latency: the relation of an edge is wave-uniform, so it selects the accumulator pair by a scalar branch
    //      (two adds executed per edge instead of an 8-way select chain), the edge loop stays rolled (two edges per
    //      trip; the code of this kernel runs once per launch from a cold instruction cache), and the mean is
    //      sum * rcp(count) with one Newton step (3 instructions; IEEE division is ~25).
    if (X || p.two_spk) {     // (the split modes exist for two-speaker graphs only: the host checks)
        // Two speakers (the reference's GNN(n_speakers = 2), cogmen.py:62-64): the relation of an edge is
        // 4 spk(source) + 2 spk(target) + (source < target ? 0 : 1) and the sources of a target are a contiguous run of rows,
        // so the four non-empty relation sums of a target are DIFFERENCES of per-speaker prefix sums over the tile's rows: a
        // row costs 6 LDS reads and 4 subtractions per lane instead of an 11-edge gather.  The prefix sums are fp64: the
        // difference, rounded to fp32, is then the (almost always correctly rounded) window sum whatever the tile's first row
        // is -- a halo row gets bit-identical means in every tile that recomputes it, which the backward relies on (it reads
        // the OWNER tile's QKVS).  Scan: 4 segments of 9 rows in parallel (800 threads), then the segment offsets.
        {
            double* const sTot = X ? reinterpret_cast<double*>(lds + XM::SM_OFF + 2048) : sRed;   // (behind plane 0's zero slot)   // [4][200] segment totals (the BatchNorm scratch / the M area is free here)
            const int sg = tid / (2 * CG_F), bc = tid % (2 * CG_F), b = bc / CG_F, c = bc % CG_F;
            double* const pf = sPfx + (b * (CG_OUT + 1) + 9 * sg) * CG_F + c;
            double acc = 0.0;
            if (tid < 8 * CG_F) {
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    pf[i * CG_F] = acc;
                    acc += sSpk[9 * sg + i] == b ? (double)sH0[(9 * sg + i) * CG_F + c] : 0.0;
                }
                sTot[sg * 2 * CG_F + bc] = acc;
            }
            __syncthreads();
            if (tid < 8 * CG_F) {
                double off = 0.0;
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) off += s2 < sg ? sTot[s2 * 2 * CG_F + bc] : 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) pf[i * CG_F] += off;
                if (sg == 3) pf[9 * CG_F] = off + acc;        // entry 36: everything
            }
        }
        __syncthreads();
        const bool act = lane < CG_F / 2;
        const int c2 = 2 * min(lane, CG_F / 2 - 1);
        // speaker masks of the outer rows (bit e: row e has speaker b), wave-uniform
        const int my_spk = lane < CG_OUT ? sSpk[lane] : -1;
        const unsigned long long mask0 = __ballot(my_spk == 0), mask1 = __ballot(my_spk == 1);
#pragma unroll 1
        for (int e = w; e < CG_MID; e += CG_NW) {
            const int node = mb + e;
            const bool valid = node >= 0 && node < N;
            const int e0 = __builtin_amdgcn_readfirstlane(min(max(sIp[e] - E_lo, 0), FW_ECAP - 1));
            const int nwin = __builtin_amdgcn_readfirstlane(valid ? min(max(sIp[e + 1] - sIp[e], 0), CG_CH) : 0);
            const int te = e + CG_HL;                                                        // outer row of the target
            const int lo = __builtin_amdgcn_readfirstlane(min(max(sSrc[e0] - ob, 0), te));   // first / one-past-last source row
            const int hi1 = nwin > 0 ? min(lo + nwin, CG_OUT) : lo;
            const int a = __builtin_amdgcn_readfirstlane(max(sSpk[te], 0));                  // target speaker
            const int past_hi = min(te, hi1);        // sources [lo, te) are "past" (direction bit 0), [te, hi1) the rest
            auto range_mask = [](int x0, int x1) -> unsigned long long {   // bits [x0, x1)
                return x1 > x0 ? ((~0ull >> (64 - (x1 - x0))) << x0) : 0ull;
            };
            const unsigned long long mp = range_mask(lo, past_hi), mf = range_mask(max(te, lo), hi1);
            int cnt[4];   // group 2 b + dir
            cnt[0] = __popcll(mask0 & mp), cnt[1] = __popcll(mask0 & mf), cnt[2] = __popcll(mask1 & mp), cnt[3] = __popcll(mask1 & mf);
            const double* const p0 = sPfx + c2, * const p1 = sPfx + (CG_OUT + 1) * CG_F + c2;
            const double2 a_lo = *reinterpret_cast<const double2*>(p0 + lo * CG_F), a_te = *reinterpret_cast<const double2*>(p0 + past_hi * CG_F);
            const double2 a_hi = *reinterpret_cast<const double2*>(p0 + hi1 * CG_F);
            const double2 b_lo = *reinterpret_cast<const double2*>(p1 + lo * CG_F), b_te = *reinterpret_cast<const double2*>(p1 + past_hi * CG_F);
            const double2 b_hi = *reinterpret_cast<const double2*>(p1 + hi1 * CG_F);
            float2 sum[4];
            sum[0] = make_float2((float)(a_te.x - a_lo.x), (float)(a_te.y - a_lo.y)), sum[1] = make_float2((float)(a_hi.x - a_te.x), (float)(a_hi.y - a_te.y));
            sum[2] = make_float2((float)(b_te.x - b_lo.x), (float)(b_te.y - b_lo.y)), sum[3] = make_float2((float)(b_hi.x - b_te.x), (float)(b_hi.y - b_te.y));
            const int le = min(max(node - ob, 0), CG_OUT - 1);
            float2 self = *reinterpret_cast<const float2*>(sH0 + le * CG_F + c2);
            if (!valid) self = make_float2(0.f, 0.f);
            uint32_t pk4[4];
            float2 mq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float fc = (float)max(cnt[q], 1), rc = __builtin_amdgcn_rcpf(fc);
                float m0 = sum[q].x * rc, m1 = sum[q].y * rc;
                m0 = fmaf(fmaf(-m0, fc, sum[q].x), rc, m0), m1 = fmaf(fmaf(-m1, fc, sum[q].y), rc, m1);
                pk4[q] = cnt[q] > 0 ? ((uint32_t)f2bf(m0) | ((uint32_t)f2bf(m1) << 16)) : 0u;
                mq[q] = cnt[q] > 0 ? make_float2(m0, m1) : make_float2(0.f, 0.f);
            }
            const uint32_t pks = (uint32_t)f2bf(self.x) | ((uint32_t)f2bf(self.y) << 16);
            const bool own = valid && e >= CG_HL && e < CG_HL + CG_TR;
            if constexpr (X) {
                // compact row [4 relation means | self] x 100 as NT term planes; the tile's own rows also as the fp32 [9 x 100] row
                // of the weight-gradient operand (relation r = 4 b + 2 a + dir: group (b, dir) lands in block 2 a + {0, 1, 4, 5})
                if (act) {
                    uint32_t* const crow = reinterpret_cast<uint32_t*>(lds + XM::SM_OFF) + e * (FX_CM / 2) + lane;
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        unsigned tt[NT];
                        sp_split2<NT>(q < 4 ? mq[q & 3].x : self.x, q < 4 ? mq[q & 3].y : self.y, tt);
#pragma unroll
                        for (int t = 0; t < NT; ++t) crow[t * (FX_MPLANE / 4) + q * (CG_F / 2)] = tt[t];
                    }
                    if (own) {
                        float* const grow = p.Mf + (int64_t)node * p.ldmf + c2;
#pragma unroll
                        for (int q = 0; q < CG_R; ++q) {
                            const int grp = (q >> 2) * 2 + (q & 1);
                            *reinterpret_cast<float2*>(grow + q * CG_F) = ((q >> 1) & 1) == a ? mq[grp] : make_float2(0.f, 0.f);
                        }
                        *reinterpret_cast<float2*>(grow + CG_R * CG_F) = self;
                    }
                }
            }
            uint32_t* const mrow = reinterpret_cast<uint32_t*>(sM + e * CG_SM);
            uint32_t* const grow = reinterpret_cast<uint32_t*>(p.Mb + (int64_t)(own ? node : 0) * p.ldmb);
            // relation r = 4 b + 2 a + dir: groups (b, dir) = 0..3 land in blocks 2 a + {0, 1, 4, 5}; the other four are empty
            if (!X && act) {
#pragma unroll
                for (int q = 0; q < CG_R; ++q) {
                    const int grp = (q >> 2) * 2 + (q & 1);            // which group block q would hold
                    const uint32_t v = ((q >> 1) & 1) == a ? pk4[grp] : 0u;
                    mrow[q * (CG_F / 2) + lane] = v;
                    if (own) grow[q * (CG_F / 2) + lane] = v;
                }
                mrow[CG_R * (CG_F / 2) + lane] = pks;
                if (own) grow[CG_R * (CG_F / 2) + lane] = pks;
            }
            if (own && lane < CG_R) {
                const int grp = (lane >> 2) * 2 + (lane & 1);
                int c = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) c = grp == q ? cnt[q] : c;
                c = ((lane >> 1) & 1) == a ? c : 0;
                p.inv_cnt[(int64_t)node * CG_R + lane] = c > 0 ? 1.0f / (float)c : 0.f;
            }
        }
    } else
    {
        const bool act = lane < CG_F / 2;
        const int c2 = 2 * min(lane, CG_F / 2 - 1);
#pragma unroll 1
        for (int e = w; e < CG_MID; e += CG_NW) {
            const int node = mb + e;
            const bool valid = node >= 0 && node < N;
            const int e0 = __builtin_amdgcn_readfirstlane(max(sIp[e] - E_lo, 0));
            const int nwin = __builtin_amdgcn_readfirstlane(valid ? min(max(sIp[e + 1] - sIp[e], 0), CG_CH) : 0);
            // lane u holds edge u of the row
            const int xm = min(e0 + min(lane, max(nwin - 1, 0)), FW_ECAP - 1);
            const int my_off = min(max(sSrc[xm] - ob, 0), CG_OUT - 1) * CG_F, my_typ = lane < nwin ? sTyp[xm] : CG_R;
            float s0[CG_R], s1[CG_R];
#pragma unroll
            for (int q = 0; q < CG_R; ++q) s0[q] = s1[q] = 0.f;
            int cnt_l = 0;                        // lane q < 8: number of in-edges of relation q
#pragma unroll 1
            for (int u = 0; u < nwin; u += 2) {   // two edges per trip, the second masked on an odd tail
                const int u1 = min(u + 1, nwin - 1);
                const int o0 = __builtin_amdgcn_readlane(my_off, u), ty0 = __builtin_amdgcn_readlane(my_typ, u);
                const int o1 = __builtin_amdgcn_readlane(my_off, u1);
                const int ty1 = u + 1 < nwin ? __builtin_amdgcn_readlane(my_typ, u1) : CG_R;
                const float2 v0 = *reinterpret_cast<const float2*>(sH0 + o0 + c2);
                const float2 v1 = *reinterpret_cast<const float2*>(sH0 + o1 + c2);
                CG_ACC8(ty0, s0, s1, v0.x, v0.y);       // relation ids >= R are ignored (PyG loops over range(num_relations))
                CG_ACC8(ty1, s0, s1, v1.x, v1.y);
                cnt_l += (ty0 == lane ? 1 : 0) + (ty1 == lane ? 1 : 0);
            }
            const int le = min(max(node - ob, 0), CG_OUT - 1);
            float2 self = *reinterpret_cast<const float2*>(sH0 + le * CG_F + c2);
            if (!valid) self = make_float2(0.f, 0.f);
            const bool own = valid && e >= CG_HL && e < CG_HL + CG_TR;   // wave-uniform: the tile's own rows go to global memory too
            uint32_t* const mrow = reinterpret_cast<uint32_t*>(sM + e * CG_SM);
            uint32_t* const grow = reinterpret_cast<uint32_t*>(p.Mb + (int64_t)(own ? node : 0) * p.ldmb);
            const float fc_l = (float)max(cnt_l, 1), rc_l = __builtin_amdgcn_rcpf(fc_l);   // lane q: count and 1 / count of relation q
#pragma unroll
            for (int q = 0; q < CG_R; ++q) {
                // mean = sum / count: q0 = s * rc, one correction step with the exact residual (sums are 0 when the count is)
                const float fc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fc_l), q));
                const float rc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rc_l), q));
                float m0 = s0[q] * rc, m1 = s1[q] * rc;
                m0 = fmaf(fmaf(-m0, fc, s0[q]), rc, m0), m1 = fmaf(fmaf(-m1, fc, s1[q]), rc, m1);
                const uint32_t pk = (uint32_t)f2bf(m0) | ((uint32_t)f2bf(m1) << 16);
                if (act) {
                    mrow[q * (CG_F / 2) + lane] = pk;
                    if (own) grow[q * (CG_F / 2) + lane] = pk;
                }
            }
            const uint32_t pks = (uint32_t)f2bf(self.x) | ((uint32_t)f2bf(self.y) << 16);
            if (act) {
                mrow[CG_R * (CG_F / 2) + lane] = pks;
                if (own) grow[CG_R * (CG_F / 2) + lane] = pks;
            }
            if (own && lane < CG_R) p.inv_cnt[(int64_t)node * CG_R + lane] = cnt_l > 0 ? 1.0f / (float)cnt_l : 0.f;   // kept for the backward
        }
    }
    __syncthreads();
    CG_STAMP(2);

    // ---- stage B: H1 = M WcatT^T + b1 for rows 0..31 of the tile (rows >= 26 read the zero row); K split over the two
    //      wavefronts of a column tile, partial tiles of the second half summed through LDS
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (mma_wave) {
        if constexpr (!X) {
            const unsigned short* const a0 = sM + r * CG_SM + 8 * g + 32 * KH0 * kh;
            const unsigned short* const a1 = sM + min(16 + r, 26) * CG_SM + 8 * g + 32 * KH0 * kh;
#pragma unroll
            for (int u = 0; u < KH0; ++u) {
                if (u < nkb) {    // wave-uniform
                    const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(a0 + 32 * u);
                    const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(a1 + 32 * u);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, bx[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, bx[u], acc1, 0, 0, 0);
                }
            }
        } else {
            // lane (r, g) feeds rows r and 16 + r of the tile: speaker of each (rows >= 26 do not exist: every group reads zero)
            const int spk0 = max(sSpk[CG_HL + r], 0), spk1 = 16 + r < CG_MID ? max(sSpk[CG_HL + min(16 + r, CG_MID - 1)], 0) : 7;
            const int rb0 = r * (FX_CM * 2), rb1 = min(16 + r, CG_MID - 1) * (FX_CM * 2);
            const unsigned char* const mbase = lds + XM::SM_OFF;
            const uint2* const lut = reinterpret_cast<const uint2*>(sLut) + 4 * KH0 * kh + g;    // entry pair of (K block u, lane group g)
#pragma unroll
            for (int u = 0; u < KH0; ++u) {
                if (u < nkb) {    // wave-uniform
                    const uint2 le = lut[4 * u];
                    int ad0[2], ad1[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const uint32_t en = hh ? le.y : le.x;
                        const int cc = (int)(en & 0xffffu), need = (int)(en >> 16);
                        ad0[hh] = (need == 0 || need == 1 + spk0) ? rb0 + cc : FX_ZERO;
                        ad1[hh] = (need == 0 || need == 1 + spk1) ? rb1 + cc : FX_ZERO;
                    }
                    sp_u32x4 fa0[NT], fa1[NT], fb[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const unsigned char* const pl = mbase + t * FX_MPLANE;
                        const uint2 x0 = *reinterpret_cast<const uint2*>(pl + ad0[0]), x1 = *reinterpret_cast<const uint2*>(pl + ad0[1]);
                        const uint2 y0 = *reinterpret_cast<const uint2*>(pl + ad1[0]), y1 = *reinterpret_cast<const uint2*>(pl + ad1[1]);
                        fa0[t] = (sp_u32x4){x0.x, x0.y, x1.x, x1.y}, fa1[t] = (sp_u32x4){y0.x, y0.y, y1.x, y1.y};
                        fb[t] = bxr[u % FX_PF<NT>][t];
                        if (u + FX_PF<NT> < KH0)      // (compile time) refill the ring slot: K block u + FX_PF, clamped (a block past the end is never multiplied)
                            bxr[u % FX_PF<NT>][t] = *reinterpret_cast<const sp_u32x4*>(brow + t * p.catT_plane + 512 * min(u + FX_PF<NT>, nkb - 1));
                    }
                    acc0 = sp_mfma<NT>(fa0, fb, acc0);
                    acc1 = sp_mfma<NT>(fa1, fb, acc1);
                }
            }
        }
        if (kh) {
            float* dst = sPart + (ct * 64 + lane) * 8;
            *reinterpret_cast<f32x4*>(dst) = acc0, *reinterpret_cast<f32x4*>(dst + 4) = acc1;
        }
    }
    // B fragments of the QKVS product (wavefront w: column tiles w and w + 16 of 25; K = 128): in flight across the barrier
    // (split modes: the first column tile's NT planes; the second tile's are requested while the first is multiplied)
    bf16x8 fq[X ? 1 : 2][4];
    sp_u32x4 fqx[X ? 4 : 1][X ? NT : 1];
    float qbias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qt = min(w + 16 * j, 24);
        const unsigned short* const bq = p.Wq + ((int64_t)qt * 4 * 64 + lane) * 8;
        if constexpr (!X) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) fq[j][kb] = *reinterpret_cast<const bf16x8*>(bq + 512 * kb);
        } else if (j == 0) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int t = 0; t < NT; ++t) fqx[kb][t] = *reinterpret_cast<const sp_u32x4*>(bq + t * p.q_plane + 512 * kb);
        }
        qbias[j] = p.bq[16 * qt + r];
    }
    __syncthreads();
    if (mma_wave && !kh) {
        const float* src = sPart + (ct * 64 + lane) * 8;
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(src), p1 = *reinterpret_cast<const f32x4*>(src + 4);
        const int col = 16 * ct + r;
        const float bias = p.b1[min(col, CG_F - 1)];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 16 * h + 4 * g + i;
                const float v = (h ? acc1[i] + p1[i] : acc0[i] + p0[i]) + bias;
                const int node = mb + e;
                if constexpr (!X) {
                    const unsigned short hb = col < CG_F ? f2bf(v) : (unsigned short)0;
                    sH1[e * CG_SH1 + col] = hb;
                    if (col < CG_F && e >= CG_HL && e < CG_HL + CG_TR && node < N) p.H1b[(int64_t)node * p.ldh1b + col] = hb;
                } else {
                    float rem = col < CG_F ? v : 0.f;      // the NT terms of the value, one per plane
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const __bf16 hb = (__bf16)rem;
                        rem -= (float)hb;
                        sH1[t * (FW_SH1_BYTES / 2) + e * CG_SH1 + col] = __builtin_bit_cast(unsigned short, hb);
                    }
                    if (col < CG_F && e >= CG_HL && e < CG_HL + CG_TR && node < N) p.H1f[(int64_t)node * p.ldh1f + col] = v;
                }
            }
    } else if (w == 7) {
        // columns [112, 128) of the H1 tile are K padding of the next product
        for (int i = lane; i < 32 * 16 * NT; i += 64) sH1[(i >> 9) * (FW_SH1_BYTES / 2) + ((i >> 4) & 31) * CG_SH1 + 112 + (i & 15)] = 0;
    }
    __syncthreads();
    CG_STAMP(3);

    // ---- stage C: QKVS = H1 Wq^T + bq, 25 column tiles over the 16 wavefronts, both row tiles; K = 128 (4 blocks)
    {
        bf16x8 fa[X ? 1 : 2][4];
        if constexpr (!X) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) fa[h][kb] = *reinterpret_cast<const bf16x8*>(sH1 + (16 * h + r) * CG_SH1 + 32 * kb + 8 * g);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int qt = w + 16 * j;
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if constexpr (X) {
                // K block by K block: the A fragments of both row tiles from the NT planes of the H1 tile, the weight fragments of
                // this column tile from registers -- and, behind each block's products, the next column tile's fragments requested
                // into the registers just freed
                const unsigned short* const bq1 = p.Wq + ((int64_t)min(w + 16, 24) * 4 * 64 + lane) * 8;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    sp_u32x4 fa0[NT], fa1[NT], fb[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const unsigned short* const pl = sH1 + t * (FW_SH1_BYTES / 2) + 32 * kb + 8 * g;
                        fa0[t] = *reinterpret_cast<const sp_u32x4*>(pl + r * CG_SH1), fa1[t] = *reinterpret_cast<const sp_u32x4*>(pl + (16 + r) * CG_SH1);
                        fb[t] = fqx[kb][t];
                        if (j == 0 && NT == 2) fqx[kb][t] = *reinterpret_cast<const sp_u32x4*>(bq1 + t * p.q_plane + 512 * kb);
                    }
                    acc[0] = sp_mfma<NT>(fa0, fb, acc[0]);
                    acc[1] = sp_mfma<NT>(fa1, fb, acc[1]);
                }
                if (j == 0 && NT > 2) {      // (three planes: refilling block by block spills -- one exposed L2 latency instead)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int t = 0; t < NT; ++t) fqx[kb][t] = *reinterpret_cast<const sp_u32x4*>(bq1 + t * p.q_plane + 512 * kb);
                }
            } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h][kb], fq[j][kb], acc[h], 0, 0, 0);
            }
            if (qt < 25) {   // wave-uniform
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 16 * h + 4 * g + i;
                        if (e < CG_MID) sQ[e * CG_SQ + 16 * qt + r] = acc[h][i] + qbias[j];
                    }
            }
        }
    }
    __syncthreads();
    CG_STAMP(4);

    // ---- own rows of the QKVS tile -> global (the backward reads q, k, v of the neighbourhoods): 16 x 100 float4
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + CG_NTH * j;
        const int li = i / 100, q = i % 100;
        const int node = r0 + li;
        if (i < CG_TR * 100 && node < N)
            *reinterpret_cast<f32x4*>(p.QKVS + (int64_t)node * 400 + 4 * q) = *reinterpret_cast<const f32x4*>(sQ + (CG_HL + li) * CG_SQ + 4 * q);
    }

    // ---- stage D: attention of the 16 own rows (one per wavefront), softmax grouped by target; lane l < 50 owns
    //      channels 2l, 2l + 1
    {
        const bool act = lane < CG_F / 2;
        const int c2 = 2 * min(lane, CG_F / 2 - 1);
        const float am = act ? 1.f : 0.f;
        const int li = w, node = r0 + li, e = CG_HL + li;
        const bool valid = node < N;
        const int e0 = __builtin_amdgcn_readfirstlane(max(sIp[e] - E_lo, 0));
        const int nwin = __builtin_amdgcn_readfirstlane(valid ? min(max(sIp[e + 1] - sIp[e], 0), CG_CH) : 0);
        float2 qv = *reinterpret_cast<const float2*>(sQ + e * CG_SQ + c2);
        qv.x *= am, qv.y *= am;
        const float2 sk = *reinterpret_cast<const float2*>(sQ + e * CG_SQ + 3 * CG_F + c2);
        int ljs[CG_CH];
#pragma unroll
        for (int u = 0; u < CG_CH; ++u) ljs[u] = sSrc[min(e0 + min(u, max(nwin - 1, 0)), FW_ECAP - 1)];
        float2 kv[CG_CH], vv[CG_CH];
#pragma unroll
        for (int u = 0; u < CG_CH; ++u) {
            const float* row = sQ + min(max(ljs[u] - mb, 0), CG_MID - 1) * CG_SQ;
            kv[u] = *reinterpret_cast<const float2*>(row + CG_F + c2), vv[u] = *reinterpret_cast<const float2*>(row + 2 * CG_F + c2);
        }
        float sc[CG_CH], part[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) part[u] = u < CG_CH ? qv.x * kv[u < CG_CH ? u : 0].x + qv.y * kv[u < CG_CH ? u : 0].y : 0.f;
        wave_sums_ch(part, sc, lane);
        float mx = -INFINITY, den = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int u = 0; u < CG_CH; ++u) {
            sc[u] *= p.scale;
            if (u < nwin) mx = fmaxf(mx, sc[u]);
        }
#pragma unroll
        for (int u = 0; u < CG_CH; ++u) {
            const float pw = u < nwin ? expf(sc[u] - mx) : 0.f;
            sc[u] = pw;
            den += pw;
            o0 += pw * vv[u].x, o1 += pw * vv[u].y;
        }
        const float inv = 1.0f / (den + 1e-16f);
        const float2 y = make_float2(valid ? o0 * inv + sk.x : 0.f, valid ? o1 * inv + sk.y : 0.f);
        // invalid rows (past N) put zeros into the BatchNorm sums
        if (act) *reinterpret_cast<float2*>(sH2 + li * CG_F + c2) = y;
        if (valid) {
            if (act) *reinterpret_cast<float2*>(p.H2 + (int64_t)node * p.ldh2 + c2) = y;
            float mine = 0.f;
#pragma unroll
            for (int u = 0; u < CG_CH; ++u)
                if (lane == u) mine = sc[u];
            if (lane < nwin) p.alpha[sIp[e] + lane] = mine * inv;
        }
    }
    CG_STAMP(5);
    if (!p.bn_fused) return;   // uniform

    // ---- BatchNorm statistics (torch.nn.BatchNorm1d in training mode, cogmen.py:67): column sums of this tile in
    //      fp64, published with write-through stores; the last workgroup to arrive adds the tiles in order
    __syncthreads();
    if (tid < 2 * CG_F) {
        const int c = tid % CG_F, sq = tid / CG_F;
        double s = 0.0;
#pragma unroll
        for (int li = 0; li < CG_TR; ++li) {
            const double x = (double)sH2[li * CG_F + c];
            s += sq ? x * x : x;
        }
        if (p.bn_fused == 2) {   // the head kernel adds the tiles (erc_head_fused_bn): plain stores, no arrival
            reinterpret_cast<float*>(p.bn_part)[(int64_t)blockIdx.x * 2 * CG_F + tid] = (float)s;
        } else {
            st_sc1d(p.bn_part + (int64_t)blockIdx.x * 2 * CG_F + tid, s);
        }
    }
    if (p.bn_fused == 2) {
        CG_STAMP(6);
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CG_STAMP(6);
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(p.bn_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (int)gridDim.x - 1;
        if (last) __hip_atomic_store(p.bn_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
    {
        const int slot = tid & 255, part = tid >> 8;        // four threads per column sum: quarters of the tile list, in order
        const int G = (int)gridDim.x, Gq = (G + 3) >> 2;
        const int g_begin = min(part * Gq, G), g_end = min(G, g_begin + Gq);
        double s = 0.0;
        if (slot < 2 * CG_F) {
            for (int g0 = g_begin; g0 < g_end; g0 += 32) {
                double t[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) t[j] = ld_sc1d(p.bn_part + (int64_t)min(g0 + j, G - 1) * 2 * CG_F + slot);
#pragma unroll
                for (int j = 0; j < 32; ++j) s += t[j] * (g0 + j < g_end ? 1.0 : 0.0);
            }
        }
        sRed[part * 256 + slot] = s;
        __syncthreads();
        if (tid < CG_F) {
            const int c = tid;
            const double sx = ((sRed[c] + sRed[256 + c]) + sRed[512 + c]) + sRed[768 + c];
            const double sxx = ((sRed[CG_F + c] + sRed[256 + CG_F + c]) + sRed[512 + CG_F + c]) + sRed[768 + CG_F + c];
            const double m = sx / (double)N;
            double var = sxx / (double)N - m * m;
            if (var < 0.0) var = 0.0;
            p.saved[c] = (float)m;
            p.saved[CG_F + c] = (float)(1.0 / sqrt(var + (double)p.eps));
            const double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
            p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)m;
            p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
        }
        if (p.stamps && tid == 0) p.stamps[7] = __builtin_amdgcn_s_memrealtime();   // the last arriver, whichever tile it is
    }
}

// sum over the 16 lanes of a row (lanes 16 g .. 16 g + 15), in every lane: DPP exchanges, no LDS traffic
template <int CTRL>
__device__ __forceinline__ float cg_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float cg_row16_sum(float v) {
    v += cg_dpp<0xB1>(v), v += cg_dpp<0x4E>(v), v += cg_dpp<0x141>(v), v += cg_dpp<0x140>(v);   // quad xor 1 / xor 2, row_half_mirror, row_mirror
    return v;
}

// ------------------------------------------------------------------------------------------------------ backward
// LDS map of the backward kernel (bytes)
constexpr int BW_ROWS = 48;                                     // the far (46) / outer (36) row tiles padded to three MFMA tiles of 16 (pad rows zero)
constexpr int BW_SK_OFF = 0;                                    // K rows [r0-15, r0+31) fp32
constexpr int BW_SK_BYTES = BW_ROWS * CG_F * 4;                 // 19200
constexpr int BW_SV_OFF = BW_SK_OFF + BW_SK_BYTES;
constexpr int BW_SQ_OFF = BW_SV_OFF + BW_SK_BYTES;              // Q rows [r0-10, r0+26)
constexpr int BW_SQ_BYTES = BW_ROWS * CG_F * 4;                 // 19200
constexpr int BW_BAND = 68;                                     // pitch of the band matrices [outer row][far column] (68 mod 64 = 4: conflict-free row reads)
constexpr int BW_SG_OFF = BW_SQ_OFF + BW_SQ_BYTES;              // dH2 rows [r0-10, r0+26)
constexpr int BW_SE_OFF = BW_SG_OFF + BW_SQ_BYTES;              // edge arrays
constexpr int BW_ECAP = 400;                                    // in-edges of the 36 outer rows (<= 36 x 11)
constexpr int BW_OCAP = 288;                                    // out-edges of the 26 mid rows (<= 26 x 11)
constexpr int BW_SE_BYTES = (3 * BW_ECAP + 4 * BW_OCAP + 40 + 28) * 4;
constexpr int BW_SDQ_OFF = BW_SE_OFF + ((BW_SE_BYTES + 15) / 16) * 16;
constexpr int CG_SDQ = 424;                                     // pitch of the dQKVS tile (bf16; K padded to 416)
constexpr int BW_SDQ_BYTES = 32 * CG_SDQ * 2;                   // 27136
constexpr int BW_SDH1_OFF = BW_SDQ_OFF + BW_SDQ_BYTES;
constexpr int BW_SDH1_BYTES = 32 * CG_F * 4;                    // 12800
constexpr int BW_SDP_OFF = BW_SDH1_OFF + BW_SDH1_BYTES;         // the dP tile (bf16 [16][968])
constexpr int BW_SDP_BYTES = CG_TR * 968 * 2;                   // 30976
constexpr int BW_LDS = BW_SDP_OFF + BW_SDP_BYTES;
constexpr int CG_KB = 936;                                      // K of the dH0 product: 9 relation blocks of 104
constexpr int CG_KBP = 960;                                     // padded to 30 blocks of 32
constexpr int CG_SDP = 968;                                     // pitch of the dP tile (bf16)
static_assert(CG_TR * CG_SDP * 2 <= BW_SDP_BYTES && BW_SDP_OFF % 16 == 0 && BW_LDS <= 160 * 1024, "backward LDS map");
static_assert(2 * BW_ROWS * BW_BAND * 4 <= BW_SDP_BYTES, "the two band matrices live in the dP tile's area until the dP stage");
static_assert(3 * 4 * 2 * 64 * 4 * 4 <= BW_SDQ_BYTES, "K-split partials of the band product live in the dQKVS tile's area");

// ---- split compute modes: LDS map of the backward kernel.  The dQKVS tile is NT planes of [27][424] (26 mid rows + one zero row),
// the dP tile NT COMPACT planes [16][5 x 104]: a source with speaker b only has out-edges of relations 4 b .. 4 b + 3 (two-speaker
// graphs), so its row of dP holds four relation blocks + self; the dH0 product runs over the K = 960 of the weight shadow through a
// lookup table (as the forward's H1 product).  Regions are ordered so that the dQKVS planes grow from the V tile (dead after the
// band product dA) into the area behind it.
constexpr int BX_DQ_ROWS = CG_MID + 1;                           // 27
constexpr int BX_DQPLANE = BX_DQ_ROWS * CG_SDQ * 2;              // 22896 bytes per term plane
constexpr int BX_CP = 528;                                       // pitch of the compact dP tile (bf16 elements)
constexpr int BX_DPPLANE = CG_TR * BX_CP * 2;                    // 16896
constexpr int BX_ZERO = 2 * 520;                                 // plane-relative byte offset of 8 zero bytes in a dP plane (row 0, columns [520, 524))
template <int NT> constexpr int BX_PF = 3;                       // weight fragments requested ahead (K blocks; see FX_PF)
constexpr int BX_SE_BYTES16 = ((BW_SE_BYTES + 15) / 16) * 16;
template <int NT>
struct BxMap {
    static constexpr int SK_OFF = 0, SQ_OFF = BW_SK_BYTES, SG_OFF = 2 * BW_SK_BYTES;      // K | Q | G row tiles; stage 3: K-split partials; stages 4, 5: dP planes
    static constexpr int SE_OFF = 3 * BW_SK_BYTES;
    static constexpr int BAND_OFF = SE_OFF + BX_SE_BYTES16;                                // band matrices; stage 3 on: the dH1 tile | K-split partials of stage 5
    static constexpr int BAND_BYTES = 2 * BW_ROWS * BW_BAND * 4;
    static constexpr int SV_OFF = BAND_OFF + BAND_BYTES;                                   // V row tile; from (c) on: the dQKVS planes
    static constexpr int E_OFF = SV_OFF + BW_SK_BYTES;                                     // head records / K-split partials of the band product; dQKVS planes
    static constexpr int E_BYTES = NT * BX_DQPLANE - BW_SK_BYTES > 24576 ? NT * BX_DQPLANE - BW_SK_BYTES : 24576;
    static constexpr int LUT_OFF = E_OFF + E_BYTES;
    static constexpr int LDS = LUT_OFF + 1024;
    static_assert(NT * BX_DPPLANE <= 3 * BW_SK_BYTES && BW_SDH1_BYTES + 7 * 64 * 4 * 4 <= BAND_BYTES && LDS <= 160 * 1024 && SV_OFF % 16 == 0 &&
                      E_OFF % 16 == 0, "split backward LDS map");
};

struct CgBwdP {
    const float* dY;               // [N, F]: dL/d(BatchNorm output) (through the LeakyReLU), from the head kernel
    const float* H2;               // [N, ldh2]: BatchNorm input
    const float* gamma;
    const float* saved;            // [0,F) mean, [F,2F) rstd
    const float* bn_bwd;           // [0,F) mean of dY, [F,2F) mean of dY * xhat
    const float* QKVS;             // [N, 400]
    const float* alpha;            // [E]
    const int32_t* in_ptr;
    const int32_t* in_src;
    const int32_t* out_ptr;
    const int32_t* out_dst;
    const int32_t* out_typ;
    const int32_t* out_eid;
    const float* inv_cnt;          // [N, 8]
    const int32_t* node_spk;       // speakers; two_spk: all in {0, 1}
    // head_part != null: the head kernel left its workgroup records un-reduced (erc_head_fused_bn, defer_reduce): every
    // workgroup here adds them up (same order everywhere) instead of reading bn_bwd; workgroup 0 publishes the results
    const float* head_part;        // [head_parts][hp_floats]: column sums of dY | of dY * xhat | loss, hits, weight sum
    float* bn_bwd_out;             // [2F]
    float* dgamma;
    float* dbeta;
    float* stats;
    int head_parts, hp_floats;
    const unsigned short* WqT;     // bf16, fragment order (7 x 13 x 512)
    const unsigned short* Wb;      // bf16, fragment order (7 x 30 x 512)
    float* dQKVS;                  // out [N, 400]      (grads_bf16: the three are bf16 buffers, pitches 400 / lddh1 / lddh0
    float* dH1;                    // out [N, lddh1]     elements -- they are only ever operands of the bf16 weight-gradient
    float* dH0;                    // out [N, lddh0]     products, csrc/wgrad_bf16.hip; pad columns are left untouched)
    float scale;
    int N, ldh2, lddh0, lddh1, grads_bf16, two_spk;
    const int32_t* n_dev;          // capacity mode: true node count (see CgFwdP)
    uint64_t* stamps;
    int stamp_block;
    int64_t qT_plane, wb_plane;    // split compute modes: WqT / Wb are NT term planes, this many elements apart
};

template <int NT>
__global__ __launch_bounds__(CG_NTH) void cogmen_bwd_tile_kernel(const CgBwdP p) {
    constexpr bool X = NT > 1;
    using XM = BxMap<X ? NT : 2>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* const sK = reinterpret_cast<float*>(lds + (X ? XM::SK_OFF : BW_SK_OFF));
    float* const sV = reinterpret_cast<float*>(lds + (X ? XM::SV_OFF : BW_SV_OFF));
    float* const sQq = reinterpret_cast<float*>(lds + (X ? XM::SQ_OFF : BW_SQ_OFF));
    float* const sG = reinterpret_cast<float*>(lds + (X ? XM::SG_OFF : BW_SG_OFF));
    float* const sAl = reinterpret_cast<float*>(lds + (X ? XM::SE_OFF : BW_SE_OFF));          // alpha of in-edge E_lo + i
    int* const sSrc = reinterpret_cast<int*>(sAl + 2 * BW_ECAP);           // its source node
    int* const sOd = sSrc + BW_ECAP;                                       // out-edge O_lo + i: target node
    int* const sOt = sOd + 2 * BW_OCAP;                                    //   relation
    float* const sOw = reinterpret_cast<float*>(sOt + BW_OCAP);            //   1 / count of (target, relation)
    int* const sIp = reinterpret_cast<int*>(sOw + BW_OCAP);                // in_ptr of outer rows (37)
    int* const sOp = sIp + 40;                                             // out_ptr of mid rows (27)
    int* const sSpkOwn = reinterpret_cast<int*>(sAl + BW_ECAP);            // speakers of the 16 own rows (the unused d(score) slot)
    unsigned short* const sDQ = reinterpret_cast<unsigned short*>(lds + (X ? XM::SV_OFF : BW_SDQ_OFF));
    float* const sDH1 = reinterpret_cast<float*>(lds + (X ? XM::BAND_OFF : BW_SDH1_OFF));
    unsigned short* const sDP = reinterpret_cast<unsigned short*>(lds + (X ? XM::SK_OFF : BW_SDP_OFF));
    float* const sPart3 = sK;                                              // K-split partials of the dH1 product
    float* const sPart5 = X ? reinterpret_cast<float*>(lds + XM::BAND_OFF + BW_SDH1_BYTES) : sG;   // K-split partials of the dH0 product
    uint32_t* const sLut = reinterpret_cast<uint32_t*>(lds + XM::LUT_OFF);                     // (split modes)
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too: scalar branches below
    const int N = p.n_dev ? min(max(*p.n_dev, 1), p.N) : p.N;   // (uniform scalar load; rows >= N of a capacity-sized grid are masked)
    const int r0 = (int)blockIdx.x * CG_TR;
    const int mb = r0 - CG_HL, ob = r0 - 2 * CG_HL, fb = r0 - 3 * CG_HL;
    const int c1 = min(lane + 64, CG_F - 1);
    const bool h1 = lane + 64 < CG_F;
    const int ct = w & 7, kh = w >> 3;           // MFMA stages: column tile and K half of this wavefront
    const bool mma_wave = ct < CG_NT;
    CG_STAMP(0);

    // ---- stage 0: graph slices and row tiles -> LDS
    const int o_lo_node = min(max(ob, 0), N), o_hi_node = min(max(ob + CG_OUT, 0), N);
    const int m_lo_node = min(max(mb, 0), N), m_hi_node = min(max(mb + CG_MID, 0), N);
    const int E_lo = p.in_ptr[o_lo_node], E_hi = min(p.in_ptr[o_hi_node], E_lo + BW_ECAP);
    const int O_lo = p.out_ptr[m_lo_node], O_hi = min(p.out_ptr[m_hi_node], O_lo + BW_OCAP);
    if (tid < CG_OUT + 1) sIp[tid] = p.in_ptr[min(max(ob + tid, 0), N)];
    if (tid >= 64 && tid < 64 + CG_MID + 1) sOp[tid - 64] = p.out_ptr[min(max(mb + tid - 64, 0), N)];
    if (tid >= 128 && tid < 128 + CG_TR) sSpkOwn[tid - 128] = p.node_spk[min(r0 + tid - 128, N - 1)];
    double* const sHp = reinterpret_cast<double*>(lds + (X ? XM::E_OFF : BW_SDQ_OFF));                  // [4][256] partial sums of the head records
    float* const sBnB = reinterpret_cast<float*>(lds + (X ? XM::E_OFF : BW_SDQ_OFF) + 4 * 256 * 8);     // [224] mean dY | mean dY * xhat
    if constexpr (X) {
        // the K -> compact column table of the dH0 product (one entry per 4 k of the K = 960 = 9 blocks of 104 + padding): compact
        // BYTE offset | needed source speaker + 1 << 16 (0: any -- the self block; 3: none)
        if (tid >= 768 && tid < 1024) {
            const int k4 = 4 * (tid - 768), q = k4 / 104, within = k4 - q * 104;
            uint32_t e = 3u << 16;
            if (q < CG_R) e = (uint32_t)(2 * ((q & 3) * 104 + within)) | ((1u + (uint32_t)(q >> 2)) << 16);
            else if (q == CG_R) e = (uint32_t)(2 * (4 * 104 + within));
            sLut[tid - 768] = e;
        }
    }
    if (p.head_part) {   // (uniform) the head's workgroup records: slot = tid % 256, a quarter of the record list each
        const int slot = tid & 255, part = tid >> 8;
        const int G = p.head_parts, Gq = (G + 3) >> 2;
        const int g_begin = min(part * Gq, G), g_end = min(G, g_begin + Gq);
        double acc = 0.0;
        if (slot < p.hp_floats) {
            for (int g0 = g_begin; g0 < g_end; g0 += 16) {
                float t[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) t[j] = p.head_part[(int64_t)min(g0 + j, G - 1) * p.hp_floats + slot];
#pragma unroll
                for (int j = 0; j < 16; ++j) acc += (double)t[j] * (g0 + j < g_end ? 1.0 : 0.0);
            }
        }
        sHp[part * 256 + slot] = acc;
    }
    {
        // K / V rows of the far range: 46 (+ 2 zero) rows x 50 float4
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + CG_NTH * j;
            const int f = min(i / 50, BW_ROWS - 1), q = i % 50;
            const int node = fb + f;
            const bool ok = f < CG_FAR && node >= 0 && node < N;
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.QKVS + (int64_t)min(max(node, 0), N - 1) * 400 + CG_F + 4 * q);
            const float m = ok ? 1.f : 0.f;
            float* dst = (q < 25 ? sK + f * CG_F + 4 * q : sV + f * CG_F + 4 * (q - 25));
            if (i < BW_ROWS * 50) *reinterpret_cast<f32x4*>(dst) = (f32x4){v.x * m, v.y * m, v.z * m, v.w * m};
        }
        // edge slices
        if (tid < BW_ECAP) {
            const int ei = min(E_lo + tid, max(E_hi - 1, E_lo));
            const bool ok = E_lo + tid < E_hi;
            const float a = p.alpha[ei];
            const int sj = p.in_src[ei];
            sAl[tid] = ok ? a : 0.f;
            sSrc[tid] = ok ? sj : 0;
        }
        if (tid >= 512 && tid < 512 + BW_OCAP) {
            const int i = tid - 512;
            const int oi = min(O_lo + i, max(O_hi - 1, O_lo));
            const bool ok = O_lo + i < O_hi;
            const int dst = p.out_dst[oi], typ = p.out_typ[oi];
            const float wgt = p.inv_cnt[(int64_t)dst * CG_R + min(typ, CG_R - 1)];
            sOd[i] = ok ? dst : 0;
            sOt[i] = ok ? typ : CG_R;
            sOw[i] = (ok && typ < CG_R) ? wgt : 0.f;
        }
    }
    // Q rows and the operands of dH2 (BatchNorm backward) for the outer range: 36 (+ 12 zero) rows x 25 float4, requested
    // BEFORE the wait for the head records below
    f32x4 r_q[2], r_dy[2], r_x[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + CG_NTH * j;
        const int e = min(i / 25, BW_ROWS - 1), q = i % 25;
        const int64_t nc = min(max(ob + e, 0), N - 1);
        r_q[j] = *reinterpret_cast<const f32x4*>(p.QKVS + nc * 400 + 4 * q);
        r_dy[j] = *reinterpret_cast<const f32x4*>(p.dY + nc * CG_F + 4 * q);
        r_x[j] = *reinterpret_cast<const f32x4*>(p.H2 + nc * p.ldh2 + 4 * q);
    }
    if (p.head_part) {
        __syncthreads();
        if (tid < 228) {
            const double sv = ((sHp[tid] + sHp[256 + tid]) + sHp[512 + tid]) + sHp[768 + tid];
            if (tid < 224) sBnB[tid] = (float)(sv / (double)N);
            if (blockIdx.x == 0) {   // dbeta = sum dY, dgamma = sum dY * xhat (BatchNorm1d backward), loss / accuracy statistics
                const int c = tid < 112 ? tid : tid - 112;
                if (tid < 112 && c < CG_F) p.dbeta[c] = (float)sv, p.bn_bwd_out[c] = (float)(sv / (double)N);
                if (tid >= 112 && tid < 224 && c < CG_F) p.dgamma[c] = (float)sv, p.bn_bwd_out[CG_F + c] = (float)(sv / (double)N);
                const double wsum = ((sHp[226] + sHp[256 + 226]) + sHp[512 + 226]) + sHp[768 + 226];
                const double wmean = wsum / (double)p.head_parts;      // every record carries the same weight sum
                if (tid == 224) p.stats[0] = (float)(sv / wmean), p.stats[2] = (float)wmean;
                if (tid == 225) p.stats[1] = (float)sv;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + CG_NTH * j;
        const int e = min(i / 25, BW_ROWS - 1), q = i % 25;
        const int node = ob + e;
        const bool ok = e < CG_OUT && node >= 0 && node < N;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(p.saved + 4 * q), rs = *reinterpret_cast<const f32x4*>(p.saved + CG_F + 4 * q);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + 4 * q);
        const f32x4 ma = p.head_part ? *reinterpret_cast<const f32x4*>(sBnB + 4 * q) : *reinterpret_cast<const f32x4*>(p.bn_bwd + 4 * q);
        const f32x4 mbv = p.head_part ? *reinterpret_cast<const f32x4*>(sBnB + 112 + 4 * q) : *reinterpret_cast<const f32x4*>(p.bn_bwd + CG_F + 4 * q);
        const float m = ok ? 1.f : 0.f;
        f32x4 gq, qq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            gq[t] = ga[t] * rs[t] * (r_dy[j][t] - ma[t] - (r_x[j][t] - mu[t]) * rs[t] * mbv[t]) * m;
            qq[t] = r_q[j][t] * m;
        }
        if (i < BW_ROWS * 25) {
            *reinterpret_cast<f32x4*>(sG + e * CG_F + 4 * q) = gq;
            *reinterpret_cast<f32x4*>(sQq + e * CG_F + 4 * q) = qq;
        }
    }
    __syncthreads();
    CG_STAMP(1);

    // ---- stages 1 / 2: the attention backward as BAND products on the matrix cores (v_mfma_f32_16x16x4_f32: exact
    //      fp32, a k-ordered fma chain).  A target only sees sources within +-5 rows, so the (target, source) quantities
    //      of a 16-target row tile live in two 16-column tiles of the far range:
    //        dA = G V^T (band)  ->  ds = alpha (dA - sum_s alpha dA) scale  (elementwise on the band, DPP row sums)
    //        dq = DS K,  dk = DS^T Q,  dv = AL^T G                          (band x row tiles, K = 32 sources / targets)
    //      One wavefront per graph row with 12-edge gathers cost 9.5 us for these two stages (wave-instruction bound).
    float* const sDS = reinterpret_cast<float*>(lds + (X ? XM::BAND_OFF : BW_SDP_OFF));            // [48][68] d(score), 0 outside the band
    float* const sAL = sDS + BW_ROWS * BW_BAND;                               // [48][68] alpha in the same layout
    float* const sPartA = reinterpret_cast<float*>(lds + (X ? XM::E_OFF : BW_SDQ_OFF));         // K-split partials (the dQKVS tile is written later)
    {
        // (a) dA: row tile i = w >> 2 (outer rows 16 i ..), K quarter j = w & 3 (25 k-steps as 7 + 7 + 7 + 4), both
        //     column tiles (far rows 16 i .. 16 i + 31); wavefronts 12..15 zero the band matrices meanwhile
        const int bi = w >> 2, bj = w & 3;
        f32x4 dA0 = {0.f, 0.f, 0.f, 0.f}, dA1 = {0.f, 0.f, 0.f, 0.f};
        if (bi < 3) {
            const float* const ga = sG + (16 * bi + r) * CG_F + g;
            const float* const v0 = sV + min(16 * bi + r, BW_ROWS - 1) * CG_F + g;
            const float* const v1 = sV + min(16 * (bi + 1) + r, BW_ROWS - 1) * CG_F + g;
            const int ks0 = 7 * bj, ks1 = min(25, ks0 + 7);
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int ks = min(ks0 + u, 24);
                const float m = ks0 + u < ks1 ? 1.f : 0.f;
                const float af = ga[4 * ks] * m;
                dA0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, v0[4 * ks], dA0, 0, 0, 0);
                dA1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, v1[4 * ks], dA1, 0, 0, 0);
            }
            float* dst = sPartA + (((bi * 4 + bj) * 2) * 64 + lane) * 4;
            *reinterpret_cast<f32x4*>(dst) = dA0, *reinterpret_cast<f32x4*>(dst + 256) = dA1;
        } else {
            for (int i = lane + 64 * (w - 12); i < 2 * BW_ROWS * BW_BAND / 4; i += 256) reinterpret_cast<f32x4*>(sDS)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        // (b) the band elementwise: the four wavefronts of a row tile each take one of the four target rows a lane holds
        //     (q = K quarter): lane (r, g) -> target 16 i + 4 g + q x sources 16 (i + c) + r (c < 2)
        if (bi < 3) {
            const int q = bj;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float* src = sPartA + (((bi * 4 + jj) * 2) * 64 + lane) * 4 + q;
                a0 += src[0], a1 += src[256];
            }
            const int t = 16 * bi + 4 * g + q;                 // outer row
            const int node_t = ob + t;
            const bool tv = t < CG_OUT && node_t >= 0 && node_t < N;
            const int tc = min(t, CG_OUT - 1);
            const int e0 = min(max(sIp[tc] - E_lo, 0), BW_ECAP - 1);
            const int cnt = tv ? min(max(sIp[tc + 1] - sIp[tc], 0), CG_CH) : 0;
            const int lo = sSrc[e0];                           // first source node of the target's in-edges
            const int o0 = (fb + 16 * bi + r) - lo, o1 = o0 + 16;   // edge position of source column r of the two tiles
            const bool k0 = o0 >= 0 && o0 < cnt, k1 = o1 >= 0 && o1 < cnt;
            const float al0 = k0 ? sAl[min(e0 + max(o0, 0), BW_ECAP - 1)] : 0.f;
            const float al1 = k1 ? sAl[min(e0 + max(o1, 0), BW_ECAP - 1)] : 0.f;
            const float x0 = k0 ? al0 * a0 : 0.f, x1 = k1 ? al1 * a1 : 0.f;
            const float tsum = cg_row16_sum(x0 + x1);
            const float d0 = k0 ? al0 * (a0 - tsum) * p.scale : 0.f, d1 = k1 ? al1 * (a1 - tsum) * p.scale : 0.f;
            float* drow = sDS + t * BW_BAND + 16 * bi + r;
            float* arow = sAL + t * BW_BAND + 16 * bi + r;
            drow[0] = d0, drow[16] = d1;
            arow[0] = al0, arow[16] = al1;
        }
        __syncthreads();
    }
    CG_STAMP(2);

    // B fragments of the dH1 product (K = 416: blocks [7 kh, 7 kh + 7), the second half has 6): requested now
    const unsigned short* const brow3 = p.WqT + ((int64_t)(min(ct, CG_NT - 1) * 13 + 7 * kh) * 64 + lane) * 8;
    bf16x8 fb3[X ? 1 : 7];
    sp_u32x4 fb3r[X ? BX_PF<NT> : 1][X ? NT : 1], bxr[X ? BX_PF<NT> : 1][X ? NT : 1];      // split modes: rings of BX_PF K blocks x NT planes
    if constexpr (!X) {
#pragma unroll
        for (int u = 0; u < 7; ++u) fb3[u] = *reinterpret_cast<const bf16x8*>(brow3 + 512 * min(u, kh ? 5 : 6));
    } else {
#pragma unroll
        for (int u = 0; u < BX_PF<NT>; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) fb3r[u][t] = *reinterpret_cast<const sp_u32x4*>(brow3 + t * p.qT_plane + 512 * min(u, kh ? 5 : 6));
    }

    const unsigned short* const brow5 = p.Wb + ((int64_t)(min(ct, CG_NT - 1) * 30 + 15 * kh) * 64 + lane) * 8;
    bf16x8 bx[X ? 1 : 15];
    {
        // (c) the dQKVS tile [26 mid rows][400] (bf16, the A operand of the dH1 product; own rows also to global memory):
        //     K padding / dead rows, the skip part (= dH2), then 42 units of 8 MFMAs: dq for outer row tiles 0, 1 (7 column
        //     tiles each), dk and dv for mid row tiles 0, 1
        uint32_t* const sDQw = reinterpret_cast<uint32_t*>(sDQ);
        constexpr int DQPW = BX_DQPLANE / 4;                  // dwords per plane (split modes)
        if constexpr (!X) {
            if (tid < CG_MID * 12) sDQw[(tid / 12) * (CG_SDQ / 2) + 200 + (tid % 12)] = 0u;
            for (int i = tid; i < 6 * (CG_SDQ / 2); i += CG_NTH) sDQw[CG_MID * (CG_SDQ / 2) + i] = 0u;
        } else {
            // K padding [400, 424) of the 26 rows and the whole zero row 26, in every plane
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (tid < CG_MID * 12) sDQw[t * DQPW + (tid / 12) * (CG_SDQ / 2) + 200 + (tid % 12)] = 0u;
                if (tid >= 512 && tid < 512 + CG_SDQ / 2) sDQw[t * DQPW + CG_MID * (CG_SDQ / 2) + tid - 512] = 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + CG_NTH * j;                 // 26 rows x 50 channel pairs
            const int em = i / 50, cp = i % 50;
            if (i < CG_MID * 50) {
                const float2 gv = *reinterpret_cast<const float2*>(sG + (em + CG_HL) * CG_F + 2 * cp);
                if constexpr (!X) {
                    sDQw[em * (CG_SDQ / 2) + 150 + cp] = (uint32_t)f2bf(gv.x) | ((uint32_t)f2bf(gv.y) << 16);
                } else {
                    unsigned tt[NT];
                    sp_split2<NT>(gv.x, gv.y, tt);
#pragma unroll
                    for (int t = 0; t < NT; ++t) sDQw[t * DQPW + em * (CG_SDQ / 2) + 150 + cp] = tt[t];
                }
                const int node = mb + em;
                if (em >= CG_HL && em < CG_HL + CG_TR && node < N) {
                    if (p.grads_bf16) *reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned short*>(p.dQKVS) + (int64_t)node * 400 + 3 * CG_F + 2 * cp) = (uint32_t)f2bf(gv.x) | ((uint32_t)f2bf(gv.y) << 16);
                    else *reinterpret_cast<float2*>(p.dQKVS + (int64_t)node * 400 + 3 * CG_F + 2 * cp) = gv;
                }
            }
        }
        // (the B fragments of the dH0 product, K = 960: blocks [15 kh, 15 kh + 15), are requested five at a time between the
        //  units: a burst of 15 loads in front of the transposed-means stage stalled it by ~3 us)
#pragma unroll
        for (int ui = 0; ui < 3; ++ui) {
            if constexpr (!X) {
#pragma unroll
                for (int t5 = 0; t5 < 5; ++t5) bx[5 * ui + t5] = *reinterpret_cast<const bf16x8*>(brow5 + 512 * (5 * ui + t5));
            }
            const int u = w + CG_NW * ui;
            if (u >= 42) continue;      // wave-uniform
            // unit -> (kind: 0 dq | 1 dk | 2 dv, row tile, column tile)
            const int kind = u < 14 ? 0 : 1 + (u - 14) / 14, uu = u < 14 ? u : (u - 14) % 14;
            const int rt = uu / 7, cti = uu % 7;
            const int col = min(16 * cti + r, CG_F - 1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (kind == 0) {       // dq[t][c] = sum_s DS[t][s] K[s][c]: t = outer row 16 rt + r, s = far rows 16 rt .. + 31
                const float* const a = sDS + (16 * rt + r) * BW_BAND + 16 * rt + g;
                const float* const bq = sK + (16 * rt + g) * CG_F + col;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * ks], bq[4 * ks * CG_F], acc, 0, 0, 0);   // rows <= 16 + 3 + 28 = 47: inside the padded tile
            } else {               // dk[i][c] = sum_t DS[t][i] Q[t][c], dv[i][c] = sum_t AL[t][i] G[t][c]: i = mid row 16 rt + r
                                   // (far column 16 rt + r + 10), t = outer rows 16 rt .. + 31
                const float* const a = (kind == 1 ? sDS : sAL) + (16 * rt + g) * BW_BAND + 16 * rt + r + 2 * CG_HL;
                const float* const bq = (kind == 1 ? sQq : sG) + (16 * rt + g) * CG_F + col;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * ks * BW_BAND], bq[4 * ks * CG_F], acc, 0, 0, 0);
            }
            // result rows 4 g + q of the tile, column 16 cti + r: into the dQKVS tile (bf16) and, for own rows, to global
            const int coff = kind * CG_F;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int em = kind == 0 ? 16 * rt + 4 * g + q - CG_HL : 16 * rt + 4 * g + q;   // mid row
                const int node = mb + em;
                const bool rowv = em >= 0 && em < CG_MID;
                if (rowv && 16 * cti + r < CG_F) {
                    const bool nv = node >= 0 && node < N;
                    if constexpr (!X) {
                        sDQ[em * CG_SDQ + coff + 16 * cti + r] = f2bf(nv ? acc[q] : 0.f);
                    } else {
                        float rem = nv ? acc[q] : 0.f;      // the NT terms of the value, one per plane
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const __bf16 hb = (__bf16)rem;
                            rem -= (float)hb;
                            sDQ[t * (BX_DQPLANE / 2) + em * CG_SDQ + coff + 16 * cti + r] = __builtin_bit_cast(unsigned short, hb);
                        }
                    }
                    if (nv && em >= CG_HL && em < CG_HL + CG_TR) {
                        if (p.grads_bf16) reinterpret_cast<unsigned short*>(p.dQKVS)[(int64_t)node * 400 + coff + 16 * cti + r] = f2bf(acc[q]);
                        else p.dQKVS[(int64_t)node * 400 + coff + 16 * cti + r] = acc[q];
                    }
                }
            }
        }
    }
    __syncthreads();
    CG_STAMP(3);

    // ---- stage 3: dH1 = dQKVS WqT^T (K = 416: 13 blocks split 7 + 6 over the two wavefronts of a column tile), both row tiles
    {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (mma_wave) {
            const unsigned short* const a0 = sDQ + r * CG_SDQ + 8 * g + 32 * 7 * kh;
            if constexpr (!X) {
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                if (u < (kh ? 6 : 7)) {   // wave-uniform
                    const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(a0 + 32 * u);
                    const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(a0 + 16 * CG_SDQ + 32 * u);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, fb3[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, fb3[u], acc1, 0, 0, 0);
                }
            }
            } else {
                const int nkb3 = kh ? 6 : 7;
                const unsigned short* const a1 = sDQ + min(16 + r, CG_MID) * CG_SDQ + 8 * g + 32 * 7 * kh;      // rows >= 26: the zero row
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    if (u < nkb3) {   // wave-uniform
                        sp_u32x4 fa0[NT], fa1[NT], fb[NT];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            fa0[t] = *reinterpret_cast<const sp_u32x4*>(a0 + t * (BX_DQPLANE / 2) + 32 * u);
                            fa1[t] = *reinterpret_cast<const sp_u32x4*>(a1 + t * (BX_DQPLANE / 2) + 32 * u);
                            fb[t] = fb3r[u % BX_PF<NT>][t];
                            if (u + BX_PF<NT> < 7) fb3r[u % BX_PF<NT>][t] = *reinterpret_cast<const sp_u32x4*>(brow3 + t * p.qT_plane + 512 * min(u + BX_PF<NT>, nkb3 - 1));
                        }
                        acc0 = sp_mfma<NT>(fa0, fb, acc0);
                        acc1 = sp_mfma<NT>(fa1, fb, acc1);
                    }
                }
                // the first blocks of the dH0 product's weights, into the registers the ring above has just freed: in flight across
                // the transposed-means stage
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < BX_PF<NT>; ++u)
#pragma unroll
                    for (int t = 0; t < NT; ++t) bxr[u][t] = *reinterpret_cast<const sp_u32x4*>(brow5 + t * p.wb_plane + 512 * u);
            }
            if (kh) {
                float* dst = sPart3 + (ct * 64 + lane) * 8;
                *reinterpret_cast<f32x4*>(dst) = acc0, *reinterpret_cast<f32x4*>(dst + 4) = acc1;
            }
        }
        __syncthreads();
        if (mma_wave && !kh) {
            const float* src = sPart3 + (ct * 64 + lane) * 8;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(src), p1 = *reinterpret_cast<const f32x4*>(src + 4);
            const int col = 16 * ct + r;
            if (col < CG_F) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 16 * h + 4 * g + i;
                        const float v = h ? acc1[i] + p1[i] : acc0[i] + p0[i];
                        sDH1[e * CG_F + col] = v;
                        const int node = mb + e;
                        if (e >= CG_HL && e < CG_HL + CG_TR && node < N) {
                            if (p.grads_bf16) reinterpret_cast<unsigned short*>(p.dH1)[(int64_t)node * p.lddh1 + col] = f2bf(v);
                            else p.dH1[(int64_t)node * p.lddh1 + col] = v;
                        }
                    }
            }
        }
    }
    __syncthreads();
    CG_STAMP(4);

    // ---- stage 4: dP = transposed relation means of dH1 for the 16 own rows (one per wavefront): block r of row j
    //      = sum over out-edges (j -> i, relation r) of dH1[i] / count_r(i); block 8 = dH1[j].  Scalar-branch
    //      accumulation and a rolled edge loop as in the forward aggregation.
    {
        const bool act = lane < CG_F / 2;                 // lane l < 50 owns channels 2l, 2l + 1
        const int c2 = 2 * min(lane, CG_F / 2 - 1);
        uint32_t* const sDPw = reinterpret_cast<uint32_t*>(sDP);
        const int li = w, em = CG_HL + li, node = r0 + li;
        const bool valid = node < N;
        const int o0 = __builtin_amdgcn_readfirstlane(sOp[em] - O_lo), o1 = __builtin_amdgcn_readfirstlane(sOp[em + 1] - O_lo);
        const int nwin = valid ? min(max(o1 - o0, 0), CG_CH) : 0;
        const int xb = max(o0, 0);
        // lane u holds out-edge u of the row
        const int xm = min(xb + min(lane, max(nwin - 1, 0)), BW_OCAP - 1);
        const int my_off = min(max(sOd[xm] - mb, 0), CG_MID - 1) * CG_F, my_typ = lane < nwin ? sOt[xm] : CG_R;
        const float my_w = sOw[xm];
        float s0[CG_R], s1[CG_R];
#pragma unroll
        for (int q = 0; q < CG_R; ++q) s0[q] = s1[q] = 0.f;
        float t0[4] = {0.f, 0.f, 0.f, 0.f}, t1[4] = {0.f, 0.f, 0.f, 0.f};      // (two-speaker path) the four non-empty relation blocks
        int b4 = 0;
        if (X || p.two_spk) {
            // two speakers: an out-edge of a source with speaker b has relation 4 b + (2 spk(target) + dir), so only the
            // four blocks 4 b .. 4 b + 3 of the row are non-empty: all 12 target rows are requested at once and a 4-way
            // scalar branch picks the accumulator (the rolled two-edges-per-trip loop below pays an LDS round trip per trip)
            b4 = 4 * __builtin_amdgcn_readfirstlane(min(max(sSpkOwn[li], 0), 1));
            float2 v[CG_CH];
            float wv[CG_CH];
            int gid[CG_CH];
#pragma unroll
            for (int u = 0; u < CG_CH; ++u) {
                const int uu = min(u, max(nwin - 1, 0));
                const int f = __builtin_amdgcn_readlane(my_off, uu);
                gid[u] = u < nwin ? __builtin_amdgcn_readlane(my_typ, uu) - b4 : -1;
                wv[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_w), uu));
                v[u] = *reinterpret_cast<const float2*>(sDH1 + f + c2);
            }
#pragma unroll
            for (int u = 0; u < CG_CH; ++u) {
                const float a0 = v[u].x * wv[u], a1 = v[u].y * wv[u];
                switch (gid[u]) {
                    case 0: t0[0] += a0, t1[0] += a1; break;
                    case 1: t0[1] += a0, t1[1] += a1; break;
                    case 2: t0[2] += a0, t1[2] += a1; break;
                    case 3: t0[3] += a0, t1[3] += a1; break;
                    default: break;
                }
            }
#pragma unroll
            for (int q = 0; q < CG_R; ++q) {
                const bool mine = (q & 4) == b4;
                s0[q] = mine ? t0[q & 3] : 0.f, s1[q] = mine ? t1[q & 3] : 0.f;
            }
        } else {
#pragma unroll 1
        for (int u = 0; u < nwin; u += 2) {
            const int u1 = min(u + 1, nwin - 1);
            const int f0 = __builtin_amdgcn_readlane(my_off, u), ty0 = __builtin_amdgcn_readlane(my_typ, u);
            const int f1 = __builtin_amdgcn_readlane(my_off, u1);
            const int ty1 = u + 1 < nwin ? __builtin_amdgcn_readlane(my_typ, u1) : CG_R;
            const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_w), u));
            const float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_w), u1));
            const float2 v0 = *reinterpret_cast<const float2*>(sDH1 + f0 + c2);
            const float2 v1 = *reinterpret_cast<const float2*>(sDH1 + f1 + c2);
            const float a0 = v0.x * w0, b0 = v0.y * w0, a1 = v1.x * w1, b1 = v1.y * w1;
            CG_ACC8(ty0, s0, s1, a0, b0);
            CG_ACC8(ty1, s0, s1, a1, b1);
        }
        }
        float2 self = *reinterpret_cast<const float2*>(sDH1 + em * CG_F + c2);
        if (!valid) self = make_float2(0.f, 0.f);
        if constexpr (!X) {
            uint32_t* const row = sDPw + li * (CG_SDP / 2);
            if (act) {
#pragma unroll
                for (int q = 0; q < CG_R; ++q) row[q * 52 + lane] = (uint32_t)f2bf(s0[q]) | ((uint32_t)f2bf(s1[q]) << 16);
                row[CG_R * 52 + lane] = (uint32_t)f2bf(self.x) | ((uint32_t)f2bf(self.y) << 16);
            }
            // padding: columns [100, 104) of every block and [936, 968)
            if (lane < CG_R + 1) row[lane * 52 + 50] = 0u, row[lane * 52 + 51] = 0u;
            if (lane >= 32 && lane < 48) row[CG_KB / 2 + (lane - 32)] = 0u;
        } else {
            // compact row [relation blocks 4 b .. 4 b + 3 | self] x 104 (columns [100, 104) zero) as NT term planes; the zero slot
            uint32_t* const row = sDPw + li * (BX_CP / 2);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                unsigned tt[NT];
                sp_split2<NT>(q < 4 ? t0[q & 3] : self.x, q < 4 ? t1[q & 3] : self.y, tt);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (act) row[t * (BX_DPPLANE / 4) + q * 52 + lane] = tt[t];
                    else if (lane < 52) row[t * (BX_DPPLANE / 4) + q * 52 + lane] = 0u;
                }
            }
            if (li == 0 && lane < 2 * NT) sDPw[(lane >> 1) * (BX_DPPLANE / 4) + BX_ZERO / 4 + (lane & 1)] = 0u;
        }
    }
    __syncthreads();
    CG_STAMP(5);

    // ---- stage 5: dH0 = dP Wb^T (K = 960: 30 blocks, 15 per wavefront of a column tile), one row tile
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (mma_wave) {
            if constexpr (!X) {
                const unsigned short* const arow = sDP + r * CG_SDP + 8 * g + 32 * 15 * kh;
#pragma unroll
                for (int u = 0; u < 15; ++u) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(arow + 32 * u);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, bx[u], acc, 0, 0, 0);
                }
            } else {
                const int spk_r = min(max(sSpkOwn[r], 0), 1);
                const int rb = r * (BX_CP * 2);
                const unsigned char* const pbase = reinterpret_cast<const unsigned char*>(sDP);
                const uint2* const lut = reinterpret_cast<const uint2*>(sLut) + 4 * 15 * kh + g;      // entry pair of (K block u, lane group g)
#pragma unroll
                for (int u = 0; u < 15; ++u) {
                    const uint2 le = lut[4 * u];
                    int ad[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const uint32_t en = hh ? le.y : le.x;
                        const int cc = (int)(en & 0xffffu), need = (int)(en >> 16);
                        ad[hh] = (need == 0 || need == 1 + spk_r) ? rb + cc : BX_ZERO;
                    }
                    sp_u32x4 fa[NT], fb[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const unsigned char* const pl = pbase + t * BX_DPPLANE;
                        const uint2 x0 = *reinterpret_cast<const uint2*>(pl + ad[0]), x1 = *reinterpret_cast<const uint2*>(pl + ad[1]);
                        fa[t] = (sp_u32x4){x0.x, x0.y, x1.x, x1.y};
                        fb[t] = bxr[u % BX_PF<NT>][t];
                        if (u + BX_PF<NT> < 15) bxr[u % BX_PF<NT>][t] = *reinterpret_cast<const sp_u32x4*>(brow5 + t * p.wb_plane + 512 * (u + BX_PF<NT>));
                    }
                    acc = sp_mfma<NT>(fa, fb, acc);
                }
            }
            if (kh) *reinterpret_cast<f32x4*>(sPart5 + (ct * 64 + lane) * 4) = acc;
        }
        __syncthreads();
        if (mma_wave && !kh) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(sPart5 + (ct * 64 + lane) * 4);
            const int col = 16 * ct + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int node = r0 + 4 * g + i;
                if (col < CG_F && node < N) {
                    if (p.grads_bf16) reinterpret_cast<unsigned short*>(p.dH0)[(int64_t)node * p.lddh0 + col] = f2bf(acc[i] + p0[i]);
                    else p.dH0[(int64_t)node * p.lddh0 + col] = acc[i] + p0[i];
                }
            }
        }
    }
    CG_STAMP(6);
}

bool ensure_lds(const void* kernel, int bytes) {
    static const void* known[4];
    static int n_known = 0;
    for (int i = 0; i < n_known; ++i)
        if (known[i] == kernel) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (n_known < 4) known[n_known++] = kernel;
    return true;
}

uint64_t* g_cg_stamps = nullptr;

}  // namespace

// diagnostic: phase stamps of the middle workgroup of the next erc_cogmen_{fwd,bwd}_tile launches, 8 x uint64 (10 ns
// ticks; tools/cogmen_stamps.py); nullptr switches them off
extern "C" int erc_cogmen_set_stamps(uint64_t* stamps) {
    g_cg_stamps = stamps;
    return ERC_OK;
}

extern "C" int64_t erc_cogmen_fwd_tile_ws_doubles(int n_nodes) { return (int64_t)erc_cdiv(n_nodes, CG_TR) * 2 * CG_F + 2; }

static int fwd_tile_launch(int terms, const float* H0, int ldh0, int n_nodes, int wp, int wf, const int32_t* in_ptr,
                           const int32_t* in_src, const int32_t* in_typ, const void* WcatT, int64_t catT_plane, const float* b1,
                           const void* Wq, int64_t q_plane, const float* bq, float scale, void* Mb, int ldmb, float* inv_cnt,
                           void* H1b, int ldh1b, float* QKVS, float* H2, int ldh2, float* alpha, int bn_fused,
                           float* running_mean, float* running_var, float momentum, float eps, float* saved,
                           double* bn_ws, const int32_t* node_spk, int n_speakers, const int32_t* n_dev, int32_t* health,
                           int32_t* events, void* stream) {
    ERC_REQUIRE(H0 && in_ptr && in_src && in_typ && WcatT && b1 && Wq && bq && Mb && inv_cnt && H1b && QKVS && H2 && alpha && node_spk,
                "cogmen_fwd_tile: null pointer");
    ERC_REQUIRE(!health == !events, "cogmen_fwd_tile: health and events come together");
    ERC_REQUIRE(n_nodes > 0 && wp >= 0 && wf >= 0 && wp <= CG_HL && wf <= CG_HL, "cogmen_fwd_tile: window (%d, %d) exceeds the halo %d",
                wp, wf, CG_HL);
    ERC_REQUIRE(ldh0 >= CG_F && ldh0 % 4 == 0 && ((uintptr_t)H0 & 15) == 0 && ldmb >= CG_KM && ldmb % 2 == 0 &&
                    ((uintptr_t)Mb & (terms > 1 ? 7 : 3)) == 0 && ldh1b >= CG_F &&
                    ldh2 >= CG_F && ldh2 % 2 == 0 && ((uintptr_t)H2 & 7) == 0 &&
                    ((uintptr_t)QKVS & 15) == 0 && ((uintptr_t)WcatT & 15) == 0 && ((uintptr_t)Wq & 15) == 0,
                "cogmen_fwd_tile: pitch / alignment");
    ERC_REQUIRE(terms == 1 || ((terms == 2 || terms == 3) && n_speakers == 2 && catT_plane > 0 && catT_plane % 8 == 0 && q_plane > 0 &&
                               q_plane % 8 == 0),
                "cogmen_fwd_tile_x: terms = %d (2 | 3), two-speaker graphs only (n_speakers = %d), plane strides multiples of 8", terms, n_speakers);
    ERC_REQUIRE(bn_fused >= 0 && bn_fused <= 2 && (bn_fused != 1 || (running_mean && running_var && saved)) && (!bn_fused || bn_ws),
                "cogmen_fwd_tile: BatchNorm operands");
    const void* kern = terms == 1 ? reinterpret_cast<const void*>(cogmen_fwd_tile_kernel<1>)
                     : terms == 2 ? reinterpret_cast<const void*>(cogmen_fwd_tile_kernel<2>)
                                  : reinterpret_cast<const void*>(cogmen_fwd_tile_kernel<3>);
    const int lds_bytes = terms == 1 ? FW_LDS : terms == 2 ? FxMap<2>::LDS : FxMap<3>::LDS;
    ERC_REQUIRE(ensure_lds(kern, lds_bytes), "cogmen_fwd_tile: %d bytes of LDS refused", lds_bytes);
    CgFwdP p{};
    p.H0 = H0; p.in_ptr = in_ptr; p.in_src = in_src; p.in_typ = in_typ; p.WcatT = (const unsigned short*)WcatT; p.b1 = b1;
    p.Wq = (const unsigned short*)Wq; p.bq = bq; p.inv_cnt = inv_cnt;
    if (terms == 1) {
        p.Mb = (unsigned short*)Mb; p.H1b = (unsigned short*)H1b; p.ldmb = ldmb; p.ldh1b = ldh1b;
    } else {
        p.Mf = (float*)Mb; p.H1f = (float*)H1b; p.ldmf = ldmb; p.ldh1f = ldh1b; p.catT_plane = catT_plane; p.q_plane = q_plane;
    }
    p.QKVS = QKVS; p.H2 = H2; p.alpha = alpha;
    const int tiles = erc_cdiv(n_nodes, CG_TR);
    p.bn_part = bn_ws ? bn_ws + 2 : nullptr;                     // [0] holds the arrival counter (8-byte slot)
    p.bn_counter = reinterpret_cast<int*>(bn_ws);
    p.running_mean = running_mean; p.running_var = running_var; p.saved = saved;
    p.momentum = momentum; p.eps = eps; p.scale = scale;
    p.N = n_nodes; p.ldh0 = ldh0; p.ldh2 = ldh2; p.bn_fused = bn_fused;
    p.node_spk = node_spk; p.two_spk = n_speakers == 2 ? 1 : 0; p.n_dev = n_dev;
    p.health = health, p.events = events;
    p.stamps = g_cg_stamps; p.stamp_block = tiles / 2;
    if (terms == 1) hipLaunchKernelGGL(cogmen_fwd_tile_kernel<1>, dim3(tiles), dim3(CG_NTH), lds_bytes, (hipStream_t)stream, p);
    else if (terms == 2) hipLaunchKernelGGL(cogmen_fwd_tile_kernel<2>, dim3(tiles), dim3(CG_NTH), lds_bytes, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(cogmen_fwd_tile_kernel<3>, dim3(tiles), dim3(CG_NTH), lds_bytes, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("cogmen_fwd_tile");
    return ERC_OK;
}

extern "C" int erc_cogmen_fwd_tile(const float* H0, int ldh0, int n_nodes, int wp, int wf, const int32_t* in_ptr,
                                   const int32_t* in_src, const int32_t* in_typ, const void* WcatT, const float* b1,
                                   const void* Wq, const float* bq, float scale, void* Mb, int ldmb, float* inv_cnt, void* H1b,
                                   int ldh1b, float* QKVS, float* H2, int ldh2, float* alpha, int bn_fused,
                                   float* running_mean, float* running_var, float momentum, float eps, float* saved,
                                   double* bn_ws, const int32_t* node_spk, int n_speakers, const int32_t* n_dev, int32_t* health,
                                   int32_t* events, void* stream) {
    return fwd_tile_launch(1, H0, ldh0, n_nodes, wp, wf, in_ptr, in_src, in_typ, WcatT, 0, b1, Wq, 0, bq, scale, Mb, ldmb, inv_cnt, H1b,
                           ldh1b, QKVS, H2, ldh2, alpha, bn_fused, running_mean, running_var, momentum, eps, saved, bn_ws, node_spk,
                           n_speakers, n_dev, health, events, stream);
}

// Split compute modes (terms = 2 | 3): WcatT / Wq = that many bf16 term planes (ErcShadowTab mode 1, planes catT_plane / q_plane
// elements apart); the weight-gradient operands are written as FP32: Mf [N, ldmf >= 900] = [mean_r H0 | H0], H1f [N, ldh1f >= 100].
// Two-speaker graphs only (n_speakers == 2).  Everything else as erc_cogmen_fwd_tile.
extern "C" int erc_cogmen_fwd_tile_x(int terms, const float* H0, int ldh0, int n_nodes, int wp, int wf, const int32_t* in_ptr,
                                     const int32_t* in_src, const int32_t* in_typ, const void* WcatT, int64_t catT_plane,
                                     const float* b1, const void* Wq, int64_t q_plane, const float* bq, float scale, float* Mf,
                                     int ldmf, float* inv_cnt, float* H1f, int ldh1f, float* QKVS, float* H2, int ldh2,
                                     float* alpha, int bn_fused, float* running_mean, float* running_var, float momentum,
                                     float eps, float* saved, double* bn_ws, const int32_t* node_spk, int n_speakers,
                                     const int32_t* n_dev, int32_t* health, int32_t* events, void* stream) {
    ERC_REQUIRE(terms == 2 || terms == 3, "cogmen_fwd_tile_x: terms = %d (2 or 3)", terms);
    return fwd_tile_launch(terms, H0, ldh0, n_nodes, wp, wf, in_ptr, in_src, in_typ, WcatT, catT_plane, b1, Wq, q_plane, bq, scale, Mf,
                           ldmf, inv_cnt, H1f, ldh1f, QKVS, H2, ldh2, alpha, bn_fused, running_mean, running_var, momentum, eps, saved,
                           bn_ws, node_spk, n_speakers, n_dev, health, events, stream);
}

static int bwd_tile_launch(int terms, const float* dY, const float* H2, int ldh2, int n_nodes, int wp, int wf, const float* gamma,
                           const float* saved, const float* bn_bwd, const float* QKVS, const float* alpha,
                           const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                           const int32_t* out_typ, const int32_t* out_eid, const float* inv_cnt, const void* WqT, int64_t qT_plane,
                           const void* Wb, int64_t wb_plane, float scale, void* dQKVS, void* dH1, void* dH0, int lddh0,
                           const int32_t* node_spk, int n_speakers, const float* head_part, int head_parts,
                           int head_part_floats, float* dgamma, float* dbeta, float* stats, int grads_bf16, int lddh1,
                           const int32_t* n_dev, void* stream) {
    ERC_REQUIRE(!head_part || (head_parts > 0 && head_part_floats >= 227 && head_part_floats <= 256 && dgamma && dbeta && stats),
                "cogmen_bwd_tile: head record operands");
    ERC_REQUIRE(dY && H2 && gamma && saved && bn_bwd && QKVS && alpha && in_ptr && in_src && out_ptr && out_dst && out_typ &&
                    out_eid && inv_cnt && WqT && Wb && dQKVS && dH1 && dH0 && node_spk, "cogmen_bwd_tile: null pointer");
    ERC_REQUIRE(n_nodes > 0 && wp >= 0 && wf >= 0 && wp <= CG_HL && wf <= CG_HL, "cogmen_bwd_tile: window (%d, %d) exceeds the halo %d",
                wp, wf, CG_HL);
    ERC_REQUIRE(lddh1 >= CG_F && (!grads_bf16 || (lddh0 % 2 == 0 && lddh1 % 2 == 0 && ((uintptr_t)dQKVS & 3) == 0)),
                "cogmen_bwd_tile: gradient pitches");
    ERC_REQUIRE(ldh2 >= CG_F && ldh2 % 4 == 0 && lddh0 >= CG_F &&
                    (((uintptr_t)dY | (uintptr_t)H2 | (uintptr_t)QKVS | (uintptr_t)gamma | (uintptr_t)saved | (uintptr_t)bn_bwd |
                      (uintptr_t)WqT | (uintptr_t)Wb) & 15) == 0, "cogmen_bwd_tile: pitch / alignment");
    ERC_REQUIRE(terms == 1 || ((terms == 2 || terms == 3) && n_speakers == 2 && !grads_bf16 && qT_plane > 0 && qT_plane % 8 == 0 &&
                               wb_plane > 0 && wb_plane % 8 == 0 && ((uintptr_t)dQKVS & 7) == 0),
                "cogmen_bwd_tile_x: terms = %d (2 | 3), two-speaker graphs only (n_speakers = %d), fp32 gradients, plane strides multiples of 8",
                terms, n_speakers);
    const void* kern = terms == 1 ? reinterpret_cast<const void*>(cogmen_bwd_tile_kernel<1>)
                     : terms == 2 ? reinterpret_cast<const void*>(cogmen_bwd_tile_kernel<2>)
                                  : reinterpret_cast<const void*>(cogmen_bwd_tile_kernel<3>);
    const int lds_bytes = terms == 1 ? BW_LDS : terms == 2 ? BxMap<2>::LDS : BxMap<3>::LDS;
    ERC_REQUIRE(ensure_lds(kern, lds_bytes), "cogmen_bwd_tile: %d bytes of LDS refused", lds_bytes);
    CgBwdP p{};
    p.dY = dY; p.H2 = H2; p.gamma = gamma; p.saved = saved; p.bn_bwd = bn_bwd; p.QKVS = QKVS; p.alpha = alpha;
    p.in_ptr = in_ptr; p.in_src = in_src; p.out_ptr = out_ptr; p.out_dst = out_dst; p.out_typ = out_typ; p.out_eid = out_eid;
    p.inv_cnt = inv_cnt; p.WqT = (const unsigned short*)WqT; p.Wb = (const unsigned short*)Wb;
    p.qT_plane = qT_plane; p.wb_plane = wb_plane;
    p.dQKVS = (float*)dQKVS; p.dH1 = (float*)dH1; p.dH0 = (float*)dH0; p.scale = scale; p.N = n_nodes; p.ldh2 = ldh2; p.lddh0 = lddh0;
    p.lddh1 = lddh1; p.grads_bf16 = grads_bf16;
    p.node_spk = node_spk; p.two_spk = n_speakers == 2 ? 1 : 0; p.n_dev = n_dev;
    p.head_part = head_part; p.head_parts = head_parts; p.hp_floats = head_part_floats; p.bn_bwd_out = const_cast<float*>(bn_bwd);
    p.dgamma = dgamma; p.dbeta = dbeta; p.stats = stats;
    p.stamps = g_cg_stamps; p.stamp_block = erc_cdiv(n_nodes, CG_TR) / 2;
    const dim3 grid(erc_cdiv(n_nodes, CG_TR));
    if (terms == 1) hipLaunchKernelGGL(cogmen_bwd_tile_kernel<1>, grid, dim3(CG_NTH), lds_bytes, (hipStream_t)stream, p);
    else if (terms == 2) hipLaunchKernelGGL(cogmen_bwd_tile_kernel<2>, grid, dim3(CG_NTH), lds_bytes, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(cogmen_bwd_tile_kernel<3>, grid, dim3(CG_NTH), lds_bytes, (hipStream_t)stream, p);
    ERC_LAUNCH_CHECK("cogmen_bwd_tile");
    return ERC_OK;
}

extern "C" int erc_cogmen_bwd_tile(const float* dY, const float* H2, int ldh2, int n_nodes, int wp, int wf, const float* gamma,
                                   const float* saved, const float* bn_bwd, const float* QKVS, const float* alpha,
                                   const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                                   const int32_t* out_typ, const int32_t* out_eid, const float* inv_cnt, const void* WqT,
                                   const void* Wb, float scale, void* dQKVS, void* dH1, void* dH0, int lddh0,
                                   const int32_t* node_spk, int n_speakers, const float* head_part, int head_parts,
                                   int head_part_floats, float* dgamma, float* dbeta, float* stats, int grads_bf16, int lddh1,
                                   const int32_t* n_dev, void* stream) {
    return bwd_tile_launch(1, dY, H2, ldh2, n_nodes, wp, wf, gamma, saved, bn_bwd, QKVS, alpha, in_ptr, in_src, out_ptr, out_dst, out_typ,
                           out_eid, inv_cnt, WqT, 0, Wb, 0, scale, dQKVS, dH1, dH0, lddh0, node_spk, n_speakers, head_part, head_parts,
                           head_part_floats, dgamma, dbeta, stats, grads_bf16, lddh1, n_dev, stream);
}

// Split compute modes (terms = 2 | 3): WqT / Wb = that many bf16 term planes, qT_plane / wb_plane elements apart; the three gradient
// tiles are written as FP32 (dQKVS [N, 400], dH1 [N, lddh1], dH0 [N, lddh0]: operands of erc_wgrad_split).  Two-speaker graphs only.
extern "C" int erc_cogmen_bwd_tile_x(int terms, const float* dY, const float* H2, int ldh2, int n_nodes, int wp, int wf,
                                     const float* gamma, const float* saved, const float* bn_bwd, const float* QKVS,
                                     const float* alpha, const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                                     const int32_t* out_dst, const int32_t* out_typ, const int32_t* out_eid, const float* inv_cnt,
                                     const void* WqT, int64_t qT_plane, const void* Wb, int64_t wb_plane, float scale, float* dQKVS,
                                     float* dH1, float* dH0, int lddh0, const int32_t* node_spk, int n_speakers,
                                     const float* head_part, int head_parts, int head_part_floats, float* dgamma, float* dbeta,
                                     float* stats, int lddh1, const int32_t* n_dev, void* stream) {
    ERC_REQUIRE(terms == 2 || terms == 3, "cogmen_bwd_tile_x: terms = %d (2 or 3)", terms);
    return bwd_tile_launch(terms, dY, H2, ldh2, n_nodes, wp, wf, gamma, saved, bn_bwd, QKVS, alpha, in_ptr, in_src, out_ptr, out_dst,
                           out_typ, out_eid, inv_cnt, WqT, qT_plane, Wb, wb_plane, scale, dQKVS, dH1, dH0, lddh0, node_spk, n_speakers,
                           head_part, head_parts, head_part_floats, dgamma, dbeta, stats, 0, lddh1, n_dev, stream);
}
