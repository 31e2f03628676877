// K8: MMGCN's 64-layer GCNII chain (track_mm/mmgcn_models.py:373-394, GraphConvolution.forward :27-39) as ONE
// persistent launch per direction.
//
// Reference, per layer l = 1..64 (theta = ln(lambda / l + 1), alpha = 0.1, variant: support = [hi | h0]):
//     hi = A h ;  out = theta [hi | h0] W_l + (1 - theta) ((1 - alpha) hi + alpha h0) ;  h <- dropout(relu(out))
// Round 1 ran this as 2 (forward) + 3 (backward) latency-bound launches per layer, 385 launches per step.  Here:
//   * re-association: out = A (h V_l) + h0 U_l with V_l = theta W_l[:200] + (1-theta)(1-alpha) I and
//     U_l = theta W_l[200:] + (1-theta) alpha I (erc_gcnii_chain_prep).  c_l = h0 U_l does not depend on the chain: all 64
//     are ONE GEMM before the launch; what stays on the chain per layer is a row-local 200 x 200 product and the
//     block product with the adjacency.
//   * the adjacency is block diagonal over dialogues, so a dialogue's rows never leave its own workgroups: workgroup
//     (dialogue b, modality m, part) owns 16 (or, when the batch would not fit the device, 32) utterance rows of one
//     modality block for all 64 layers, with its rows of the normalised adjacency block RESIDENT IN LDS and its h tile in
//     LDS.  The work table is derived ON THE DEVICE from the dialogue lengths (no host sync): every workgroup walks
//     node_off, so short dialogues take few workgroups and the usual batch runs with 16-row parts (no padded MFMA rows).
//     Per layer the parts of a (dialogue, modality) all-gather z = h V_l (and read the same utterances' rows of the
//     other modalities for the cross-modal entries) through global memory: write-through (sc1) stores, drained, one
//     flag per workgroup and layer, L1-bypassing 16-byte loads (MI355X_MICROARCH.md, hand-off rows handoff-flag /
//     publish-large).
//   * latency: V_{l+1} (26 KB per wavefront, register resident) and the layer's c_l values are requested while the
//     gathered rows are still in flight; all gather loads are issued before the first one is consumed; the block product
//     reads its adjacency operand with 16-byte LDS loads and runs one 16-deep k group ahead.
//   * backward: the same structure mirrored (A is symmetric): dg = dh . mask -> all-gather -> dz = A dg -> dh = dz V_l^T.
//     Everything that only meets in a sum over the layers (dV_l = h_l^T dz_l, dU_l = h0^T dg_l, dh0 = sum_l dg_l U_l^T, the
//     adjacency gradient sum_l dg_l z_l^T) is left to batched products after the launch; the chain saves z, dg, dz per layer.
// All products are v_mfma_f32_16x16x4_f32 (exact fp32).  Every workgroup of a dialogue must be resident: the host caps
// the grid by the occupancy query and runs the dialogues in several launches if needed; polls are bounded.
#include "erc_common.h"
#include <type_traits>

namespace {

constexpr int FD = 200;         // feature width (nhidden)
constexpr int KP = 208;         // FD padded to 13 groups of 16
constexpr int NL = 64;          // layers
constexpr int CNT = 512;        // threads per workgroup
constexpr int NT13 = 13;        // 16-column tiles over FD
constexpr int HP = 212;         // LDS row pitch of the h tile
constexpr int ZP = 204;         // LDS row pitch of the gathered z rows
constexpr int MAXRW = 32;       // rows per workgroup (two 16-row MFMA tiles)
constexpr int MAXT = 128;       // longest dialogue this kernel takes (LDS: MAXT x ZP gathered rows)
constexpr int SPIN_LIMIT = 4000000;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct Chain {
    const float* ADJ; int P;              // normalised adjacency blocks [B*Mo][P][P]
    const float* CR;                      // cross-modal entries [B][Mo*Mo][P]
    const int32_t* node_off;              // [B+1]
    int N, Mo, B, b0, nb, pmax, AP;       // dialogues [b0, b0 + nb); flags per (dialogue, modality); LDS pitch of the adjacency rows
    const float* W;                       // fwd: VT [NL][FD][KP] (row n, contiguous k) ; bwd: V [NL][FD][KP] (row k, contiguous n)
    const float* Call; int ldc;           // fwd: c_l = h0 U_l for all layers, [Mo*N][ldc], layer l at column (l-1) * FD
    float* HD; int64_t hd_plane;          // h planes [NL+2][Mo*N][FD]: plane l = input of layer l, plane NL+1 = output
    float* ZS; float* DG; float* DZ; int lds;   // per-layer saves [Mo*N][lds] (layer l at column (l-1) * FD): z (fwd) | dg, dz (bwd)
    const float* dHin; float* dHout;      // bwd: gradient wrt plane NL+1 [Mo*N][FD] in, wrt plane 1 out
    float* ZX;                            // exchange [2][Mo*N][FD]
    int* flags;                           // [B*Mo*pmax] one per workgroup: epoch * 128 + layer
    int* epoch;                           // [B]
    int* err;
    float drop_p, ks; const uint64_t* rng; uint64_t rng_stream0;     // dropout of layer l: stream rng_stream0 + l (as gcnii_layer_fwd)
    uint64_t* stamps;                     // diagnostic phase stamps of workgroup 0 (or nullptr)
    int spin_limit;                       // bound of a poll (SPIN_LIMIT; lowered by the timeout test)
};

#define CHAIN_STAMP(layer, slot)                                                                               \
    do {                                                                                                       \
        if (p.stamps && blockIdx.x == 0 && tid == 0) p.stamps[((layer) - 1) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

__device__ __forceinline__ int ld_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one workgroup's 64 layers; MTC = 16-row MFMA tiles compiled in (1: 16-row parts, 2: 32-row parts)
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load, which
// would serialise the register prefetches (next layer's weights, c_l) behind each barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool BWD, int MTC>
__device__ __forceinline__ void chain_body(const Chain& p, float* smem, int b, int L, int nparts, int m, int part, int RW) {
    const int tid = threadIdx.x, lane = tid & 63, q4 = lane >> 4, l15 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);               // scalar: conditions on it are uniform branches
    const int off_b = p.node_off[b];
    const int r0 = part * RW, nr = min(RW, L - r0);
    const int MT = MTC == 1 ? 1 : (nr + 15) >> 4;         // 16-row tiles in use
    const int64_t R3 = (int64_t)p.Mo * p.N;
    const int64_t row0 = (int64_t)m * p.N + off_b + r0;   // global row of this workgroup's first node
    const int AP = p.AP;
    float* hbuf = smem;                                   // [MAXRW][HP]   A operand of the row-local product / staging
    float* adj = hbuf + MAXRW * HP;                       // [MAXRW][AP]   this workgroup's rows of the adjacency block
    float* zbuf = adj + MAXRW * AP;                       // [MAXT][ZP]    the modality block's gathered rows
    float* crs = zbuf + (int64_t)MAXT * ZP;               // [2][MAXRW]    cross-modal coefficients of the own rows

    // ---- residents: adjacency rows, cross coefficients, the first h tile
    for (int x = tid; x < MAXRW * AP; x += CNT) adj[x] = 0.f;
    for (int x = tid; x < MAXRW * HP; x += CNT) hbuf[x] = 0.f;
    // the block product runs k in groups of 16: the rows L .. 16 ceil(L / 16) - 1 meet zero adjacency columns, but must hold
    // finite values themselves (0 * NaN is NaN)
    for (int x = tid; x < 16 * ZP; x += CNT)
        if (L + x / ZP < MAXT) zbuf[(L + x / ZP) * ZP + x % ZP] = 0.f;
    __syncthreads();
    {
        const float* blk = p.ADJ + ((int64_t)(b * p.Mo + m) * p.P + r0) * p.P;
        for (int x = tid; x < nr * L; x += CNT) adj[(x / L) * AP + x % L] = blk[(int64_t)(x / L) * p.P + x % L];
        for (int x = tid; x < 2 * MAXRW; x += CNT) {
            const int q = x / MAXRW, i = x % MAXRW, n = q + (q >= m ? 1 : 0);       // the other modalities, in order
            crs[x] = (n < p.Mo && i < nr) ? p.CR[((int64_t)b * p.Mo * p.Mo + m * p.Mo + n) * p.P + r0 + i] : 0.f;
        }
        const float* h0 = BWD ? p.dHin + row0 * FD : p.HD + p.hd_plane + row0 * FD;    // plane 1 / the incoming gradient
        for (int x = tid; x < nr * (FD / 4); x += CNT) {
            const int i = x / (FD / 4), c4 = x % (FD / 4);
            *reinterpret_cast<float4*>(hbuf + i * HP + 4 * c4) = *reinterpret_cast<const float4*>(h0 + (int64_t)i * FD + 4 * c4);
        }
    }
    const unsigned ep = (unsigned)p.epoch[b] + 1u;
    const int fbase = (b * p.Mo) * p.pmax;                // flags of this dialogue: [m][part]
    // buffer descriptor of the exchange (write-through stores, L1-bypassing loads: aux 16 = sc1)
    const uint64_t zx_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)p.ZX);
    const uint64_t zx_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)p.ZX >> 32));
    const __amdgpu_buffer_rsrc_t zxr = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((zx_hi << 32) | zx_lo), 0,
                                                                         (int)(2 * R3 * FD * 4), 0x00020000);
    __syncthreads();

    const int t1ok = wave + 8 < NT13;                     // this wavefront's column tiles: wave and (t1ok) wave + 8
    const int c0 = 16 * wave + l15, c1 = 16 * (t1ok ? wave + 8 : wave) + l15;

    // a layer's weight tiles -> registers.  Wl: row = output column, contiguous along the contraction; k runs in groups of 16
    // with the order 16 g + 4 (lane >> 4) + j on BOTH operands, so one 16-byte load feeds four MFMAs
    auto load_weights = [&](const float* Wl, f32x4 (&wr)[2][13]) {
        const float* w0 = Wl + (int64_t)c0 * KP + 4 * q4;
        const float* w1 = Wl + (int64_t)c1 * KP + 4 * q4;
#pragma unroll
        for (int g = 0; g < 13; ++g) {
            wr[0][g] = *reinterpret_cast<const f32x4*>(w0 + 16 * g);
            wr[1][g] = *reinterpret_cast<const f32x4*>(w1 + 16 * g);
        }
    };
    // row-local product: acc[mt][u] = src[16 mt .., :] . Wl
    auto row_local_t = [&](auto nu, const float* src, const f32x4 (&wr)[2][13], f32x4 (&acc)[MTC][2]) {
        constexpr int NU = decltype(nu)::value;
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[mt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* a0 = src + l15 * HP + 4 * q4;
#pragma unroll
        for (int g = 0; g < 13; ++g) {
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) {
                if (mt >= MT) continue;
                const f32x4 av = *reinterpret_cast<const f32x4*>(a0 + mt * 16 * HP + 16 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wr[0][g][j], acc[mt][0], 0, 0, 0);
                    if (NU == 2) acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wr[1][g][j], acc[mt][1], 0, 0, 0);
                }
            }
        }
    };
    auto row_local = [&](const float* src, const f32x4 (&wr)[2][13], f32x4 (&acc)[MTC][2]) {
        // wavefronts 5..7 have one real tile: their second slot repeats it (results dropped).  Uniform code for all
        // wavefronts is worth more than the skipped MFMAs: the busiest SIMD (wavefronts 0 and 4) carries 4 tile products anyway
        row_local_t(std::integral_constant<int, 2>{}, src, wr, acc);
    };
    // block product: acc[mt][u] = adj[16 mt .., :] . zbuf[:, tile], k in groups of 16 (same order trick), one group ahead.
    // The next row-local product's weight tiles (Wn) are requested between the k groups: the vector memory pipe takes
    // ~1.5 us to accept a layer's 170 KB of weights, which the MFMAs of this product hide
    auto block_product_t = [&](auto nu, f32x4 (&acc)[MTC][2], const float* Wn, f32x4 (&wr)[2][13]) {
        constexpr int NU = decltype(nu)::value;
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[mt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ng = (L + 15) >> 4;
        const float* ap = adj + l15 * AP + 4 * q4;
        const float* zp = zbuf + 4 * q4 * ZP;
        const float* w0 = Wn + (int64_t)c0 * KP + 4 * q4;
        const float* w1 = Wn + (int64_t)c1 * KP + 4 * q4;
        f32x4 av[MTC];
        float bv[2][4];
        auto load_group = [&](int g, f32x4 (&a)[MTC], float (&bb)[2][4]) {
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(ap + (mt < MT ? mt : 0) * 16 * AP + 16 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bb[0][j] = zp[(16 * g + j) * ZP + c0];
                if (NU == 2) bb[1][j] = zp[(16 * g + j) * ZP + c1];
            }
        };
        load_group(0, av, bv);
#pragma unroll
        for (int g = 0; g < MAXT / 16; ++g) {
            if (Wn) {
#pragma unroll
                for (int h = 2 * g; h < 2 * g + 2; ++h)
                    if (h < 13) {
                        wr[0][h] = *reinterpret_cast<const f32x4*>(w0 + 16 * h);
                        if (NU == 2) wr[1][h] = *reinterpret_cast<const f32x4*>(w1 + 16 * h);
                    }
            }
            if (g < ng) {
                f32x4 an[MTC];
                float bn[2][4];
                load_group(g + 1 < ng ? g + 1 : g, an, bn);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MTC; ++mt) {
                        if (mt >= MT) continue;
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][j], bv[0][j], acc[mt][0], 0, 0, 0);
                        if (NU == 2) acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][j], bv[1][j], acc[mt][1], 0, 0, 0);
                    }
#pragma unroll
                for (int mt = 0; mt < MTC; ++mt) av[mt] = an[mt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bv[0][j] = bn[0][j];
                    if (NU == 2) bv[1][j] = bn[1][j];
                }
            }
        }
    };
    auto block_product = [&](f32x4 (&acc)[MTC][2], const float* Wn, f32x4 (&wr)[2][13]) {
        block_product_t(std::integral_constant<int, 2>{}, acc, Wn, wr);
    };
    // accumulator tiles -> rows of an LDS buffer (C/D layout: column = lane & 15, rows 4 (lane >> 4) + r)
    auto tiles_to_lds = [&](const f32x4 (&acc)[MTC][2], float* dst, int pitch, int nrows) {
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = wave + 8 * u;
                if (mt >= MT || t >= NT13) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * mt + 4 * q4 + r, n = 16 * t + l15;
                    if (n < FD && i < nrows) dst[i * pitch + n] = acc[mt][u][r];
                }
            }
    };
    // the own rows are on their way to the exchange buffer: drain, raise the flag, wait for the dialogue's other parts of
    // this modality and for the same part of the other modalities
    auto publish_wait = [&](int l, auto&& window) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // every storing wavefront drains ...
        __syncthreads();                                                     // ... before the one flag store
        CHAIN_STAMP(l, 3);
        const int want = (int)(ep * 128u + (unsigned)l);
        if (tid == 0) st_i32(p.flags + fbase + m * p.pmax + part, want);
        window();                                          // saves that nobody waits for: issued while the flags travel
        if (wave == 0) {       // one wavefront polls: lane j < nparts -> part j of this modality, then the other modalities
            const int nwait = nparts + p.Mo - 1;
            if (lane < nwait) {
                const int* f = lane < nparts ? p.flags + fbase + m * p.pmax + lane
                                             : p.flags + fbase + ((lane - nparts) + ((lane - nparts) >= m ? 1 : 0)) * p.pmax + part;
                int spins = 0;
                while (ld_i32(f) - want < 0) {             // monotonic: a fast member may already show a later layer
                    if (++spins > p.spin_limit) {
                        st_i32(p.err, ERC_HEALTH_RAISED);
                        break;
                    }
                    if ((spins & 255) == 0 && ld_i32(p.err)) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        lds_barrier();
        CHAIN_STAMP(l, 4);
    };
    // the exchange's loads.  The modality's other rows go straight to their place in zbuf (LDS-DMA, one 800-byte row per
    // wave-instruction: no staging registers, no ds_write); the same utterances' rows of the other modalities come through
    // registers and become the weighted cross-modal sum in hbuf
    constexpr int CQ = (MTC * 16 * (FD / 4) + CNT - 1) / CNT;
    auto gather_issue = [&](int par, int tz, u32x4 (&cv)[CQ][2]) {
        const int64_t xbase = (int64_t)par * R3 * FD;
        const float* src = p.ZX + xbase + ((int64_t)m * p.N + off_b) * FD + 4 * (tz & 63);
        if ((tz & 63) < FD / 4) {
            for (int j = wave; j < L; j += CNT / 64) {
                if (j >= r0 && j < r0 + nr) continue;
                __builtin_amdgcn_global_load_lds(src + (int64_t)j * FD, (lds_void*)(zbuf + j * ZP), 16, 0, 16);
            }
        }
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const int x = tz + q * CNT, i = x / (FD / 4), c4 = x % (FD / 4);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int n = o + (o >= m ? 1 : 0);
                cv[q][o] = u32x4{0u, 0u, 0u, 0u};
                if (x < nr * (FD / 4) && o < p.Mo - 1)
                    cv[q][o] = __builtin_amdgcn_raw_buffer_load_b128(
                        zxr, (int)((xbase + ((int64_t)n * p.N + off_b + r0 + i) * FD + 4 * c4) * 4), 0, 16);
            }
        }
    };
    auto gather_consume = [&](int tz, const u32x4 (&cv)[CQ][2]) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const int x = tz + q * CNT, i = x / (FD / 4), c4 = x % (FD / 4);
            if (x < nr * (FD / 4)) {
                const f32x4 f0 = __builtin_bit_cast(f32x4, cv[q][0]), f1 = __builtin_bit_cast(f32x4, cv[q][1]);
                const float q0 = crs[i], q1 = crs[MAXRW + i];
                *reinterpret_cast<f32x4*>(hbuf + i * HP + 4 * c4) = q0 * f0 + q1 * f1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the rows written by LDS-DMA have landed
    };

    f32x4 acc[MTC][2];
    f32x4 wr[2][13];
    u32x4 cv[CQ][2];
    if (!BWD) {
        load_weights(p.W, wr);
        for (int l = 1; l <= NL; ++l) {
            // per-layer index arithmetic is recomputed from an opaque copy of the thread index: hoisted out of the layer loop
            // it would pin ~100 registers of addresses next to the register-resident weights
            int tz = tid;
            asm volatile("" : "+v"(tz));
            const int q4z = (tz & 63) >> 4, l15z = tz & 15;
            CHAIN_STAMP(l, 0);
            // z = h V_l (row-local), straight from the accumulators to the exchange buffer and the own rows of zbuf
            row_local(hbuf, wr, acc);
            CHAIN_STAMP(l, 1);
            {
                const int64_t xbase = (int64_t)(l & 1) * R3 * FD;
#pragma unroll
                for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int t = wave + 8 * u;
                        if (mt >= MT || t >= NT13) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * mt + 4 * q4z + r, n = 16 * t + l15z;
                            if (i >= nr || n >= FD) continue;
                            const float v = acc[mt][u][r];
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), zxr,
                                                                  (int)((xbase + (row0 + i) * FD + n) * 4), 0, 16);
                            zbuf[(r0 + i) * ZP + n] = v;
                        }
                    }
            }
            CHAIN_STAMP(l, 2);
            publish_wait(l, [&]() {
                // the z save and the plane this layer started from (written by the previous layer's epilogue into hbuf)
                float* zs = p.ZS + row0 * p.lds + (int64_t)(l - 1) * FD;
                float* hd = p.HD + (int64_t)l * p.hd_plane + row0 * FD;
                for (int x = tz; x < nr * (FD / 4); x += CNT) {
                    const int i = x / (FD / 4), c4 = x % (FD / 4);
                    *reinterpret_cast<f32x4*>(zs + (int64_t)i * p.lds + 4 * c4) = *reinterpret_cast<const f32x4*>(zbuf + (r0 + i) * ZP + 4 * c4);
                    if (l > 1)
                        *reinterpret_cast<f32x4*>(hd + (int64_t)i * FD + 4 * c4) = *reinterpret_cast<const f32x4*>(hbuf + i * HP + 4 * c4);
                }
            });
            gather_issue(l & 1, tz, cv);
            // this layer's c_l values, requested behind the gather
            const float* cl = p.Call + (int64_t)(l - 1) * FD;
            float cr[MTC][2][4];
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * mt + 4 * q4z + r, n = 16 * (wave + 8 * u) + l15z;
                        cr[mt][u][r] = (mt < MT && i < nr && n < FD) ? cl[(row0 + i) * p.ldc + n] : 0.f;
                    }
            CHAIN_STAMP(l, 9);
            gather_consume(tz, cv);
            CHAIN_STAMP(l, 10);
            lds_barrier();
            CHAIN_STAMP(l, 5);
            // out = A z + cross + c_l ; h' = dropout(relu(out)).  V_{l+1} is requested between the k groups
            block_product(acc, l < NL ? p.W + (int64_t)l * FD * KP : nullptr, wr);
            CHAIN_STAMP(l, 6);
            uint64_t rng_off = 0, rng_seed = 0;
            if (p.drop_p > 0.f) rng_off = p.rng[0], rng_seed = p.rng[1] ^ (p.rng_stream0 + (uint64_t)l);
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = wave + 8 * u;
                    if (mt >= MT || t >= NT13) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * mt + 4 * q4z + r, n = 16 * t + l15z;
                        if (i >= nr || n >= FD) continue;
                        float v = acc[mt][u][r] + hbuf[i * HP + n] + cr[mt][u][r];
                        v = fmaxf(v, 0.f);
                        if (p.drop_p > 0.f) {
                            const float uu = erc_uniform(rng_seed, rng_off, (uint64_t)(row0 + i) * (uint64_t)FD + n);
                            v = (uu >= p.drop_p) ? v * p.ks : 0.f;
                        }
                        hbuf[i * HP + n] = v;                                 // each element read and rewritten by its own lane
                    }
                }
            lds_barrier();
            CHAIN_STAMP(l, 7);
        }
        float* hd = p.HD + (int64_t)(NL + 1) * p.hd_plane + row0 * FD;        // the last plane (the others went out one layer late)
        for (int x = tid; x < nr * (FD / 4); x += CNT) {
            const int i = x / (FD / 4), c4 = x % (FD / 4);
            *reinterpret_cast<float4*>(hd + (int64_t)i * FD + 4 * c4) = *reinterpret_cast<const float4*>(hbuf + i * HP + 4 * c4);
        }
    } else {
        // the relu / dropout mask of layer l is the sign of plane l + 1: requested one layer ahead
        float4 mk[CQ];
        auto load_mask = [&](int l, int tz) {
            const float* hd = p.HD + (int64_t)(l + 1) * p.hd_plane + row0 * FD;
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                const int x = tz + q * CNT, i = x / (FD / 4), c4 = x % (FD / 4);
                mk[q] = x < nr * (FD / 4) ? *reinterpret_cast<const float4*>(hd + (int64_t)i * FD + 4 * c4) : float4{0.f, 0.f, 0.f, 0.f};
            }
        };
        load_mask(NL, tid);
        for (int l = NL; l >= 1; --l) {
            const int sl = NL + 1 - l;
            int tz = tid;
            asm volatile("" : "+v"(tz));
            const int q4z = (tz & 63) >> 4, l15z = tz & 15;
            CHAIN_STAMP(sl, 0);
            // dg = dh . mask(layer l's output) -> own rows of zbuf, the exchange buffer and the save
            {
                const int64_t xbase = (int64_t)(l & 1) * R3 * FD;
#pragma unroll
                for (int q = 0; q < CQ; ++q) {
                    const int x = tz + q * CNT, i = x / (FD / 4), c4 = x % (FD / 4);
                    if (x >= nr * (FD / 4)) continue;
                    const float4 o = mk[q];
                    const float4 d = *reinterpret_cast<const float4*>(hbuf + i * HP + 4 * c4);
                    f32x4 g;
                    g.x = o.x > 0.f ? d.x * p.ks : 0.f, g.y = o.y > 0.f ? d.y * p.ks : 0.f;
                    g.z = o.z > 0.f ? d.z * p.ks : 0.f, g.w = o.w > 0.f ? d.w * p.ks : 0.f;
                    *reinterpret_cast<f32x4*>(zbuf + (r0 + i) * ZP + 4 * c4) = g;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g), zxr,
                                                           (int)((xbase + (row0 + i) * FD + 4 * c4) * 4), 0, 16);
                }
            }
            CHAIN_STAMP(sl, 2);
            publish_wait(sl, [&]() {
                float* dg = p.DG + row0 * p.lds + (int64_t)(l - 1) * FD;
                for (int x = tz; x < nr * (FD / 4); x += CNT) {
                    const int i = x / (FD / 4), c4 = x % (FD / 4);
                    *reinterpret_cast<f32x4*>(dg + (int64_t)i * p.lds + 4 * c4) = *reinterpret_cast<const f32x4*>(zbuf + (r0 + i) * ZP + 4 * c4);
                }
            });
            gather_issue(l & 1, tz, cv);
            if (l > 1) load_mask(l - 1, tz);
            CHAIN_STAMP(sl, 9);
            gather_consume(tz, cv);                                           // hbuf <- the cross-modal sum (dh is consumed)
            CHAIN_STAMP(sl, 10);
            lds_barrier();
            CHAIN_STAMP(sl, 5);
            // dz = A dg + cross (A is symmetric) -> hbuf, saved.  V_l is requested between the k groups
            block_product(acc, p.W + (int64_t)(l - 1) * FD * KP, wr);
            CHAIN_STAMP(sl, 6);
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = wave + 8 * u;
                    if (mt >= MT || t >= NT13) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * mt + 4 * q4z + r, n = 16 * t + l15z;
                        if (i >= nr || n >= FD) continue;
                        hbuf[i * HP + n] += acc[mt][u][r];
                    }
                }
            lds_barrier();
            CHAIN_STAMP(sl, 7);
            for (int x = tz; x < nr * (FD / 4); x += CNT) {
                const int i = x / (FD / 4), c4 = x % (FD / 4);
                *reinterpret_cast<float4*>(p.DZ + (row0 + i) * p.lds + (int64_t)(l - 1) * FD + 4 * c4) =
                    *reinterpret_cast<const float4*>(hbuf + i * HP + 4 * c4);
            }
            // dh = dz V_l^T (row-local) -> hbuf
            row_local(hbuf, wr, acc);
            CHAIN_STAMP(sl, 8);
            lds_barrier();
            tiles_to_lds(acc, hbuf, HP, MAXRW);
            lds_barrier();
        }
        for (int x = tid; x < nr * (FD / 4); x += CNT) {
            const int i = x / (FD / 4), c4 = x % (FD / 4);
            *reinterpret_cast<float4*>(p.dHout + (row0 + i) * FD + 4 * c4) = *reinterpret_cast<const float4*>(hbuf + i * HP + 4 * c4);
        }
    }
    if (m == 0 && part == 0 && tid == 0) p.epoch[b] = (int)ep;
}

template <bool BWD>
__global__ __launch_bounds__(CNT) void gcnii_chain_kernel(Chain p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // ---- work table, from the dialogue lengths: 16-row parts if the launch's dialogues then fit the grid, else 32-row parts
    int tot16 = 0;
    for (int bb = p.b0; bb < p.b0 + p.nb; ++bb) tot16 += p.Mo * ((p.node_off[bb + 1] - p.node_off[bb] + 15) >> 4);
    const int RW = tot16 <= (int)gridDim.x ? 16 : MAXRW;
    int w = blockIdx.x, b = p.b0, L = 0, nparts = 0;
    for (;; ++b) {
        if (b >= p.b0 + p.nb) return;                     // beyond the table: nobody waits for this workgroup
        L = p.node_off[b + 1] - p.node_off[b];
        nparts = (L + RW - 1) / RW;
        if (w < p.Mo * nparts) break;
        w -= p.Mo * nparts;
    }
    const int m = w / nparts, part = w % nparts;
    if (RW == 16) chain_body<BWD, 1>(p, smem, b, L, nparts, m, part, RW);
    else chain_body<BWD, 2>(p, smem, b, L, nparts, m, part, RW);
}

// V_l = theta W_l[:200] + (1 - theta)(1 - alpha) I in both orientations, U_l = theta W_l[200:] + (1 - theta) alpha I.
// One workgroup per (32 x 32 tile, layer): the two W tiles are read once with n contiguous, V / U go out directly, VT / UT through
// an LDS transpose -- every access coalesced.  (The first version read W a second time with k contiguous -- 800-byte strides,
// cache hits -- for the transposed outputs: 52 us for 61 MB; this one is bound by those bytes.)
constexpr int PT = 32;                       // tile edge
constexpr int PNT = (FD + PT - 1) / PT;      // 7 tiles per dimension
__global__ __launch_bounds__(256) void gcnii_prep_kernel(const float* __restrict__ W, int64_t w_stride, float lamda, float alpha,
                                                         float* __restrict__ VT, float* __restrict__ V, float* __restrict__ U,
                                                         float* __restrict__ UT) {
    __shared__ float sv[PT][PT + 1], su[PT][PT + 1];
    const int l = blockIdx.y;                 // 0-based layer
    const int k0 = (blockIdx.x / PNT) * PT, n0 = (blockIdx.x % PNT) * PT;
    const float theta = logf(lamda / (float)(l + 1) + 1.f);
    const float* Wl = W + l * w_stride;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads, four rows each
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + ty + 8 * r, n = n0 + tx;
        float v = 0.f, u = 0.f;
        if (k < FD && n < FD) {
            const float d = k == n ? 1.f : 0.f;
            v = theta * Wl[k * FD + n] + (1.f - theta) * (1.f - alpha) * d;
            u = theta * Wl[(FD + k) * FD + n] + (1.f - theta) * alpha * d;
            V[((int64_t)l * FD + k) * KP + n] = v;
            U[(int64_t)k * NL * FD + l * FD + n] = u;
        }
        sv[ty + 8 * r][tx] = v, su[ty + 8 * r][tx] = u;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + ty + 8 * r, k = k0 + tx;
        if (k < FD && n < FD) {
            VT[((int64_t)l * FD + n) * KP + k] = sv[tx][ty + 8 * r];
            if (UT) UT[((int64_t)l * FD + n) * FD + k] = su[tx][ty + 8 * r];   // B operand of Call = h0 UT^T
        }
    }
    // the pad columns 200..207 of V / VT stay zero (zero-filled by the caller once)
}

int chain_ap(int T) { return 16 * ((T + 15) / 16) + 4; }      // adjacency row pitch in LDS: whole 16-groups, 16-byte rows, conflict-free
int chain_lds(int T) { return 4 * (MAXRW * HP + MAXRW * chain_ap(T) + MAXT * ZP + 2 * MAXRW); }

template <typename K>
bool chain_ensure_lds(K kernel, int lds) {
    static const void* known[4];
    static int granted[4];
    static int n_known = 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    int slot = -1;
    for (int i = 0; i < n_known; ++i)
        if (known[i] == key) slot = i;
    if (slot < 0) {
        if (n_known == 4) return false;
        slot = n_known++;
        known[slot] = key, granted[slot] = 64 * 1024;
    }
    if (lds <= granted[slot]) return true;
    if (hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return false;
    granted[slot] = lds;
    return true;
}

}  // namespace

extern "C" int erc_gcnii_chain_prep(const float* W, int64_t w_stride, float lamda, float alpha, float* VT, float* V, float* U,
                                    float* UT, void* stream) {
    ERC_REQUIRE(W && VT && V && U && w_stride >= 2 * FD * FD, "gcnii_chain_prep: bad arguments");
    hipLaunchKernelGGL(gcnii_prep_kernel, dim3(PNT * PNT, NL), dim3(256), 0, (hipStream_t)stream, W, w_stride, lamda, alpha, VT, V, U, UT);
    ERC_LAUNCH_CHECK("gcnii_chain_prep");
    return ERC_OK;
}

// The work table is built on the device; the host only fixes what it can know without the lengths: the flag pitch
// (parts = ceil(T / 16) per dialogue and modality), the grid cap (one workgroup per CU: every workgroup of a dialogue must be
// resident) and how many dialogues one launch may take in the worst case (all of them T long, 32-row parts)
extern "C" int erc_gcnii_chain_config(int B, int T, int Mo, int P, int* parts, int* grid_cap, int* dialogues_per_launch) {
    ERC_REQUIRE(B > 0 && T > 0 && T <= MAXT && Mo >= 2 && Mo <= 3 && P >= T && parts && grid_cap && dialogues_per_launch,
                "gcnii_chain_config: B=%d T=%d modalities=%d (T <= %d)", B, T, Mo, MAXT);
    int dev = 0, nf = 0, nb = 0;
    hipDeviceProp_t prop;
    ERC_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess, "gcnii_chain_config: no device");
    const int lds = chain_lds(T);
    ERC_REQUIRE(lds <= 160 * 1024, "gcnii_chain_config: %d bytes of LDS", lds);
    ERC_REQUIRE(chain_ensure_lds(gcnii_chain_kernel<false>, lds) && chain_ensure_lds(gcnii_chain_kernel<true>, lds), "gcnii_chain_config: LDS limit");
    ERC_REQUIRE(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, gcnii_chain_kernel<false>, CNT, lds) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gcnii_chain_kernel<true>, CNT, lds) == hipSuccess && nf >= 1 && nb >= 1,
                "gcnii_chain_config: occupancy query");
    const int cap = prop.multiProcessorCount;              // one workgroup per CU
    const int worst = Mo * ((T + MAXRW - 1) / MAXRW);      // workgroups of one dialogue with 32-row parts
    ERC_REQUIRE(worst <= cap, "gcnii_chain_config: a dialogue does not fit the device");
    *parts = (T + 15) / 16, *grid_cap = cap;
    const int dpl = cap / worst;
    *dialogues_per_launch = dpl < B ? dpl : B;
    return ERC_OK;
}

static uint64_t* g_chain_stamps = nullptr;
// diagnostic: phase stamps of workgroup 0, [64 layers][16] (tools/chain_stamps.py); nullptr switches them off
extern "C" int erc_gcnii_chain_set_stamps(uint64_t* stamps) {
    g_chain_stamps = stamps;
    return ERC_OK;
}
static int g_chain_spin_limit = SPIN_LIMIT;
// test hook: bound of the per-layer exchange polls (<= 0 restores the default).  With a bound of 1 the first wait that is
// not satisfied at once raises the health word: the timeout path end to end (tests/test_gpu_mmgcn.py).
extern "C" int erc_gcnii_chain_set_spin_limit(int limit) {
    g_chain_spin_limit = limit > 0 ? limit : SPIN_LIMIT;
    return ERC_OK;
}
static int chain_launch(bool bwd, const float* ADJ, int P, const float* CR, const int32_t* node_off, int N, int Mo, int B, int T,
                        int parts, int grid_cap, int dpl, const float* W, const float* Call, int ldc, float* HD, int64_t hd_plane,
                        float* ZS, float* DG, float* DZ, int lds_, const float* dHin, float* dHout, float* ZX, int32_t* state,
                        int32_t* health, float drop_p, const uint64_t* rng, uint64_t rng_stream0, void* stream) {
    const int lds = chain_lds(T);
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < B; b0 += dpl) {
        const int nb = B - b0 < dpl ? B - b0 : dpl;
        int grid = Mo * nb * ((T + 15) / 16);              // 16-row parts if the lengths allow (the device decides), never more
        if (grid > grid_cap) grid = grid_cap;
        Chain p{ADJ, P, CR, node_off, N, Mo, B, b0, nb, parts, chain_ap(T), W, Call, ldc, HD, hd_plane, ZS, DG, DZ, lds_, dHin, dHout,
                ZX, state + 1 + B, state + 1, health ? health : state, drop_p, drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, rng, rng_stream0, nullptr};
        p.stamps = b0 == 0 ? g_chain_stamps : nullptr;
        p.spin_limit = g_chain_spin_limit;
        if (bwd) {
            if (!chain_ensure_lds(gcnii_chain_kernel<true>, lds)) return ERC_E_LAUNCH;
            hipLaunchKernelGGL(gcnii_chain_kernel<true>, dim3(grid), dim3(CNT), lds, st, p);
        } else {
            if (!chain_ensure_lds(gcnii_chain_kernel<false>, lds)) return ERC_E_LAUNCH;
            hipLaunchKernelGGL(gcnii_chain_kernel<false>, dim3(grid), dim3(CNT), lds, st, p);
        }
        ERC_LAUNCH_CHECK("gcnii_chain");
    }
    return ERC_OK;
}

extern "C" int erc_gcnii_chain_fwd(const float* ADJ, int P, const float* CR, const int32_t* node_off, int N, int Mo, int B, int T,
                                   int parts, int grid_cap, int dialogues_per_launch, const float* VT, const float* Call, int ldc,
                                   float* HD, int64_t hd_plane, float* ZS, int lds, float* ZX, int32_t* state, int32_t* health,
                                   float drop_p, const uint64_t* rng_state, uint64_t rng_stream0, void* stream) {
    ERC_REQUIRE(ADJ && CR && node_off && VT && Call && HD && ZS && ZX && state, "gcnii_chain_fwd: null pointer");
    ERC_REQUIRE(N > 0 && Mo >= 2 && Mo <= 3 && B > 0 && T > 0 && T <= MAXT && P >= T && parts >= (T + 15) / 16 &&
                    dialogues_per_launch >= 1 && grid_cap >= dialogues_per_launch * Mo * ((T + MAXRW - 1) / MAXRW) &&
                    ldc >= NL * FD && lds >= NL * FD,
                "gcnii_chain_fwd: bad sizes (T=%d parts=%d grid cap=%d)", T, parts, grid_cap);
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state), "gcnii_chain_fwd: drop_p=%f", (double)drop_p);
    return chain_launch(false, ADJ, P, CR, node_off, N, Mo, B, T, parts, grid_cap, dialogues_per_launch, VT, Call, ldc, HD, hd_plane, ZS,
                        nullptr, nullptr, lds, nullptr, nullptr, ZX, state, health, drop_p, rng_state, rng_stream0, stream);
}

extern "C" int erc_gcnii_chain_bwd(const float* ADJ, int P, const float* CR, const int32_t* node_off, int N, int Mo, int B, int T,
                                   int parts, int grid_cap, int dialogues_per_launch, const float* V, const float* HD,
                                   int64_t hd_plane, const float* dHin, float* dHout, float* DG, float* DZ, int lds, float* ZX,
                                   int32_t* state, int32_t* health, float drop_p, void* stream) {
    ERC_REQUIRE(ADJ && CR && node_off && V && HD && dHin && dHout && DG && DZ && ZX && state, "gcnii_chain_bwd: null pointer");
    ERC_REQUIRE(N > 0 && Mo >= 2 && Mo <= 3 && B > 0 && T > 0 && T <= MAXT && P >= T && parts >= (T + 15) / 16 &&
                    dialogues_per_launch >= 1 && grid_cap >= dialogues_per_launch * Mo * ((T + MAXRW - 1) / MAXRW) && lds >= NL * FD,
                "gcnii_chain_bwd: bad sizes (T=%d parts=%d grid cap=%d)", T, parts, grid_cap);
    return chain_launch(true, ADJ, P, CR, node_off, N, Mo, B, T, parts, grid_cap, dialogues_per_launch, V, nullptr, 0,
                        const_cast<float*>(HD), hd_plane, nullptr, DG, DZ, lds, dHin, dHout, ZX, state, health, drop_p, nullptr, 0, stream);
}
