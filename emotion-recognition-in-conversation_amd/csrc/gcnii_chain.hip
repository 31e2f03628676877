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
//     (dialogue b, modality m, part) owns <= 32 utterance rows of one modality block for all 64 layers, with its rows of
//     the normalised adjacency block RESIDENT IN LDS (<= 32 x 110 fp32) and its h tile in LDS; V_l streams from L2 once per
//     layer and workgroup.  Per layer the parts of a (dialogue, modality) all-gather z = h V_l (and read the same
//     utterances' rows of the other modalities for the cross-modal entries) through global memory: write-through
//     (sc1) 16-byte stores, drained, one flag per workgroup and layer, L1-bypassing 16-byte loads
//     (MI355X_MICROARCH.md, hand-off rows handoff-flag / publish-large).
//   * backward: the same structure mirrored (A is symmetric): dg = dh . mask -> all-gather -> dz = A dg -> dh = dz V_l^T.
//     Everything that only meets in a sum over the layers (dV_l = h_l^T dz_l, dU_l = h0^T dg_l, dh0 = sum_l dg_l U_l^T, the
//     adjacency gradient sum_l dg_l z_l^T) is left to batched products after the launch; the chain saves z, dg, dz per layer.
// All products are v_mfma_f32_16x16x4_f32 (exact fp32).  Every workgroup of a dialogue must be resident: the host caps
// the grid by the occupancy query and runs the dialogues in several launches if needed; polls are bounded.
#include "erc_common.h"

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

struct Chain {
    const float* ADJ; int P;              // normalised adjacency blocks [B*Mo][P][P]
    const float* CR;                      // cross-modal entries [B][Mo*Mo][P]
    const int32_t* node_off;              // [B+1]
    int N, Mo, B, b0, parts, RW;          // dialogues [b0, b0 + grid / (Mo * parts)); rows per workgroup
    const float* W;                       // fwd: VT [NL][FD][KP] (row n, contiguous k) ; bwd: V [NL][FD][KP] (row k, contiguous n)
    const float* Call; int ldc;           // fwd: c_l = h0 U_l for all layers, [Mo*N][ldc], layer l at column (l-1) * FD
    float* HD; int64_t hd_plane;          // h planes [NL+2][Mo*N][FD]: plane l = input of layer l, plane NL+1 = output
    float* ZS; float* DG; float* DZ; int lds;   // per-layer saves [Mo*N][lds] (layer l at column (l-1) * FD): z (fwd) | dg, dz (bwd)
    const float* dHin; float* dHout;      // bwd: gradient wrt plane NL+1 [Mo*N][FD] in, wrt plane 1 out
    float* ZX;                            // exchange [2][Mo*N][FD]
    int* flags;                           // [B*Mo*parts] one per workgroup: epoch * 128 + layer
    int* epoch;                           // [B]
    int* err;
    float drop_p, ks; const uint64_t* rng; uint64_t rng_stream0;     // dropout of layer l: stream rng_stream0 + l (as gcnii_layer_fwd)
    uint64_t* stamps;                     // diagnostic phase stamps of workgroup 0 (or nullptr)
};

#define CHAIN_STAMP(layer, slot)                                                                               \
    do {                                                                                                       \
        if (p.stamps && blockIdx.x == 0 && tid == 0) p.stamps[((layer) - 1) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

__device__ __forceinline__ int ld_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool BWD>
__global__ __launch_bounds__(CNT) void gcnii_chain_kernel(Chain p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = blockIdx.x % p.parts, m = (blockIdx.x / p.parts) % p.Mo, b = p.b0 + blockIdx.x / (p.parts * p.Mo);
    const int off_b = p.node_off[b], L = p.node_off[b + 1] - off_b;
    const int r0 = part * p.RW, nr = min(p.RW, L - r0);
    if (nr <= 0) return;                                  // nobody waits for a part without rows
    const int MT = (nr + 15) >> 4;                        // 16-row tiles in use (<= 2)
    const int nparts = (L + p.RW - 1) / p.RW;             // parts of this dialogue that have rows
    const int64_t R3 = (int64_t)p.Mo * p.N;
    const int64_t row0 = (int64_t)m * p.N + off_b + r0;   // global row of this workgroup's first node
    float* hbuf = smem;                                   // [MAXRW][HP]   A operand of the row-local product / staging
    float* adj = hbuf + MAXRW * HP;                       // [MAXRW][AP]   this workgroup's rows of the adjacency block
    const int AP = p.P + 1;
    float* zbuf = adj + MAXRW * AP;                       // [L][ZP]       the modality block's gathered rows
    float* crs = zbuf + (int64_t)MAXT * ZP;               // [2][MAXRW]    cross-modal coefficients of the own rows
    __shared__ int s_ok;

    // ---- residents: adjacency rows, cross coefficients, the first h tile
    for (int x = tid; x < MAXRW * AP; x += CNT) adj[x] = 0.f;
    for (int x = tid; x < MAXRW * HP; x += CNT) hbuf[x] = 0.f;
    // the block product reads k in groups of 4: the (masked) rows L .. L+3 must hold finite values, 0 * NaN is NaN
    for (int x = tid; x < 4 * ZP; x += CNT)
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
    const int fbase = (b * p.Mo) * p.parts;               // flags of this dialogue: [m][part]
    // buffer descriptor of the exchange (16-byte write-through stores, L1-bypassing loads: aux 16 = sc1)
    const uint64_t zx_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)p.ZX);
    const uint64_t zx_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)p.ZX >> 32));
    const __amdgpu_buffer_rsrc_t zxr = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((zx_hi << 32) | zx_lo), 0,
                                                                         (int)(2 * R3 * FD * 4), 0x00020000);
    __syncthreads();

    // row-local product: acc[mt][u] = src[16 mt .., :] . Wl (Wl: row = output column, contiguous along the contraction);
    // a wavefront takes column tiles wave and wave + 8; k runs in groups of 16 with the order 16 g + 4 (lane >> 4) + j on BOTH
    // operands, so one 16-byte load feeds four MFMAs
    auto row_local = [&](const float* src, const float* Wl, f32x4 (&acc)[2][2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[mt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t1ok = wave + 8 < NT13;
        const float* w0 = Wl + (int64_t)(16 * wave + (lane & 15)) * KP + 4 * (lane >> 4);
        const float* w1 = Wl + (int64_t)(16 * (t1ok ? wave + 8 : wave) + (lane & 15)) * KP + 4 * (lane >> 4);
        const float* a0 = src + (lane & 15) * HP + 4 * (lane >> 4);
        for (int g0 = 0; g0 < 13; g0 += 4) {              // batches of 4 k-groups: 8 weight loads in flight
            f32x4 bw[4][2];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int g = min(g0 + gg, 12);
                bw[gg][0] = *reinterpret_cast<const f32x4*>(w0 + 16 * g);
                bw[gg][1] = *reinterpret_cast<const f32x4*>(w1 + 16 * g);
            }
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                if (g0 + gg > 12) continue;
                const int g = g0 + gg;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (mt >= MT) continue;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(a0 + mt * 16 * HP + 16 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bw[gg][0][j], acc[mt][0], 0, 0, 0);
                        if (t1ok) acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bw[gg][1][j], acc[mt][1], 0, 0, 0);
                    }
                }
            }
        }
    };
    // block product: acc[mt][u] = adj[16 mt .., :L] . zbuf[:L, tile]
    auto block_product = [&](f32x4 (&acc)[2][2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[mt][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t1ok = wave + 8 < NT13;
        const int c0 = 16 * wave + (lane & 15), c1 = 16 * (t1ok ? wave + 8 : wave) + (lane & 15);
        const int ks = (L + 3) >> 2;
        for (int s = 0; s < ks; ++s) {
            const int k = min(4 * s + (lane >> 4), MAXT - 1);
            const float kv = 4 * s + (lane >> 4) < L ? 1.f : 0.f;
            const float b0v = zbuf[k * ZP + c0] * kv, b1v = zbuf[k * ZP + c1] * kv;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                if (mt >= MT) continue;
                const float av = adj[(16 * mt + (lane & 15)) * AP + min(4 * s + (lane >> 4), p.P - 1)];
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0v, acc[mt][0], 0, 0, 0);
                if (t1ok) acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1v, acc[mt][1], 0, 0, 0);
            }
        }
    };
    // accumulator tiles -> rows of an LDS buffer (C/D layout: column = lane & 15, rows 4 (lane >> 4) + r)
    auto tiles_to_lds = [&](const f32x4 (&acc)[2][2], float* dst, int pitch, int nrows) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = wave + 8 * u;
                if (mt >= MT || t >= NT13) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * mt + 4 * (lane >> 4) + r, n = 16 * t + (lane & 15);
                    if (n < FD && i < nrows) dst[i * pitch + n] = acc[mt][u][r];
                }
            }
    };
    // publish the own rows (in zbuf at utterances r0 ..) to the exchange buffer of parity `par`, raise the flag, wait for
    // the dialogue's other parts of this modality and for the same part of the other modalities, then gather their rows:
    // the modality's rows into zbuf, the cross-modal combination sum_n cr[n][i] x_n[r0 + i, :] into hbuf
    auto exchange = [&](int l, int par, float* save, int64_t save_col, int sl) {
        const int64_t xbase = (int64_t)par * R3 * FD;
        for (int x = tid; x < nr * (FD / 4); x += CNT) {
            const int i = x / (FD / 4), c4 = x % (FD / 4);
            const f32x4 v = *reinterpret_cast<const f32x4*>(zbuf + (r0 + i) * ZP + 4 * c4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), zxr, (int)((xbase + (row0 + i) * FD + 4 * c4) * 4), 0, 16);
            if (save) *reinterpret_cast<f32x4*>(save + (row0 + i) * p.lds + save_col + 4 * c4) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // every storing wavefront drains ...
        __syncthreads();                                                     // ... before the one flag store
        CHAIN_STAMP(sl, 3);
        const int want = (int)(ep * 128u + (unsigned)l);
        if (tid == 0) st_i32(p.flags + fbase + m * p.parts + part, want);
        if (wave == 0) {       // one wavefront polls: lane j < nparts -> part j of this modality, then the other modalities
            const int nwait = nparts + p.Mo - 1;
            int ok = 1;
            if (lane < nwait) {
                const int* f = lane < nparts ? p.flags + fbase + m * p.parts + lane
                                             : p.flags + fbase + ((lane - nparts) + ((lane - nparts) >= m ? 1 : 0)) * p.parts + part;
                int spins = 0;
                while (ld_i32(f) - want < 0) {             // monotonic: a fast member may already show a later layer
                    if (++spins > SPIN_LIMIT) {
                        st_i32(p.err, 1);
                        ok = 0;
                        break;
                    }
                    if ((spins & 255) == 0 && ld_i32(p.err)) {
                        ok = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (lane == 0) s_ok = 1;
            (void)ok;
        }
        __syncthreads();
        CHAIN_STAMP(sl, 4);
        // the other parts' rows of this modality
        const int64_t mrow0 = (int64_t)m * p.N + off_b;
        for (int x = tid; x < L * (FD / 4); x += CNT) {
            const int j = x / (FD / 4), c4 = x % (FD / 4);
            if (j >= r0 && j < r0 + nr) continue;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(zxr, (int)((xbase + (mrow0 + j) * FD + 4 * c4) * 4), 0, 16);
            *reinterpret_cast<f32x4*>(zbuf + j * ZP + 4 * c4) = __builtin_bit_cast(f32x4, v);
        }
        // cross-modal rows: same utterances, other modalities, weighted
        for (int x = tid; x < nr * (FD / 4); x += CNT) {
            const int i = x / (FD / 4), c4 = x % (FD / 4);
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < p.Mo - 1; ++q) {
                const int n = q + (q >= m ? 1 : 0);
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                    zxr, (int)((xbase + ((int64_t)n * p.N + off_b + r0 + i) * FD + 4 * c4) * 4), 0, 16);
                const f32x4 f = __builtin_bit_cast(f32x4, v);
                const float cq = crs[q * MAXRW + i];
                s.x += cq * f.x, s.y += cq * f.y, s.z += cq * f.z, s.w += cq * f.w;
            }
            *reinterpret_cast<f32x4*>(hbuf + i * HP + 4 * c4) = s;
        }
        __syncthreads();
        CHAIN_STAMP(sl, 5);
    };

    f32x4 acc[2][2];
    if (!BWD) {
        for (int l = 1; l <= NL; ++l) {
            // z = h V_l (row-local) -> own rows of zbuf
            CHAIN_STAMP(l, 0);
            row_local(hbuf, p.W + (int64_t)(l - 1) * FD * KP, acc);
            CHAIN_STAMP(l, 1);
            __syncthreads();                                                  // all reads of hbuf done
            tiles_to_lds(acc, zbuf + r0 * ZP, ZP, nr);
            __syncthreads();
            CHAIN_STAMP(l, 2);
            exchange(l, l & 1, p.ZS, (int64_t)(l - 1) * FD, l);
            // out = A z + cross + c_l ; h' = dropout(relu(out))
            block_product(acc);
            CHAIN_STAMP(l, 6);
            const float* cl = p.Call + (int64_t)(l - 1) * FD;
            uint64_t rng_off = 0, rng_seed = 0;
            if (p.drop_p > 0.f) rng_off = p.rng[0], rng_seed = p.rng[1] ^ (p.rng_stream0 + (uint64_t)l);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = wave + 8 * u;
                    if (mt >= MT || t >= NT13) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * mt + 4 * (lane >> 4) + r, n = 16 * t + (lane & 15);
                        if (i >= nr || n >= FD) continue;
                        float v = acc[mt][u][r] + hbuf[i * HP + n] + cl[(row0 + i) * p.ldc + n];
                        v = fmaxf(v, 0.f);
                        if (p.drop_p > 0.f) {
                            const float uu = erc_uniform(rng_seed, rng_off, (uint64_t)(row0 + i) * (uint64_t)FD + n);
                            v = (uu >= p.drop_p) ? v * p.ks : 0.f;
                        }
                        hbuf[i * HP + n] = v;                                 // each element read and rewritten by its own lane
                    }
                }
            __syncthreads();
            CHAIN_STAMP(l, 7);
            float* hd = p.HD + (int64_t)(l + 1) * p.hd_plane + row0 * FD;     // the next plane: what the backward / the head read
            for (int x = tid; x < nr * (FD / 4); x += CNT) {
                const int i = x / (FD / 4), c4 = x % (FD / 4);
                *reinterpret_cast<float4*>(hd + (int64_t)i * FD + 4 * c4) = *reinterpret_cast<const float4*>(hbuf + i * HP + 4 * c4);
            }
            CHAIN_STAMP(l, 8);
        }
    } else {
        for (int l = NL; l >= 1; --l) {
            CHAIN_STAMP(NL + 1 - l, 0);
            // dg = dh . mask(layer l's output) -> own rows of zbuf, saved
            {
                const float* hd = p.HD + (int64_t)(l + 1) * p.hd_plane + row0 * FD;
                for (int x = tid; x < nr * (FD / 4); x += CNT) {
                    const int i = x / (FD / 4), c4 = x % (FD / 4);
                    const float4 o = *reinterpret_cast<const float4*>(hd + (int64_t)i * FD + 4 * c4);
                    const float4 d = *reinterpret_cast<const float4*>(hbuf + i * HP + 4 * c4);
                    float4 g;
                    g.x = o.x > 0.f ? d.x * p.ks : 0.f, g.y = o.y > 0.f ? d.y * p.ks : 0.f;
                    g.z = o.z > 0.f ? d.z * p.ks : 0.f, g.w = o.w > 0.f ? d.w * p.ks : 0.f;
                    *reinterpret_cast<float4*>(zbuf + (r0 + i) * ZP + 4 * c4) = g;
                }
            }
            __syncthreads();
            CHAIN_STAMP(NL + 1 - l, 2);
            exchange(NL + 1 - l, l & 1, p.DG, (int64_t)(l - 1) * FD, NL + 1 - l);
            // dz = A dg + cross (A is symmetric) -> hbuf, saved
            block_product(acc);
            CHAIN_STAMP(NL + 1 - l, 6);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = wave + 8 * u;
                    if (mt >= MT || t >= NT13) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * mt + 4 * (lane >> 4) + r, n = 16 * t + (lane & 15);
                        if (i >= nr || n >= FD) continue;
                        hbuf[i * HP + n] += acc[mt][u][r];
                    }
                }
            __syncthreads();
            for (int x = tid; x < nr * (FD / 4); x += CNT) {
                const int i = x / (FD / 4), c4 = x % (FD / 4);
                *reinterpret_cast<float4*>(p.DZ + (row0 + i) * p.lds + (int64_t)(l - 1) * FD + 4 * c4) =
                    *reinterpret_cast<const float4*>(hbuf + i * HP + 4 * c4);
            }
            CHAIN_STAMP(NL + 1 - l, 7);
            // dh = dz V_l^T (row-local) -> hbuf
            row_local(hbuf, p.W + (int64_t)(l - 1) * FD * KP, acc);
            CHAIN_STAMP(NL + 1 - l, 8);
            __syncthreads();
            tiles_to_lds(acc, hbuf, HP, MAXRW);
            __syncthreads();
        }
        for (int x = tid; x < nr * (FD / 4); x += CNT) {
            const int i = x / (FD / 4), c4 = x % (FD / 4);
            *reinterpret_cast<float4*>(p.dHout + (row0 + i) * FD + 4 * c4) = *reinterpret_cast<const float4*>(hbuf + i * HP + 4 * c4);
        }
    }
    if (m == 0 && part == 0 && tid == 0) p.epoch[b] = (int)ep;
}

// V_l = theta W_l[:200] + (1 - theta)(1 - alpha) I in both orientations, U_l = theta W_l[200:] + (1 - theta) alpha I
__global__ __launch_bounds__(256) void gcnii_prep_kernel(const float* __restrict__ W, int64_t w_stride, float lamda, float alpha,
                                                         float* __restrict__ VT, float* __restrict__ V, float* __restrict__ U) {
    const int l = blockIdx.y;                 // 0-based layer
    const float theta = logf(lamda / (float)(l + 1) + 1.f);
    const float* Wl = W + l * w_stride;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < FD * FD; x += gridDim.x * 256) {
        const int k = x / FD, n = x % FD;
        const float d = k == n ? 1.f : 0.f;
        const float v = theta * Wl[k * FD + n] + (1.f - theta) * (1.f - alpha) * d;
        V[((int64_t)l * FD + k) * KP + n] = v;
        VT[((int64_t)l * FD + n) * KP + k] = v;
        U[(int64_t)k * NL * FD + l * FD + n] = theta * Wl[(FD + k) * FD + n] + (1.f - theta) * alpha * d;
    }
    // the pad columns 200..207 of V / VT stay zero (zero-filled by the caller once)
}

int chain_lds(int P) { return 4 * (MAXRW * HP + MAXRW * (P + 1) + MAXT * ZP + 2 * MAXRW); }

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
                                    void* stream) {
    ERC_REQUIRE(W && VT && V && U && w_stride >= 2 * FD * FD, "gcnii_chain_prep: bad arguments");
    hipLaunchKernelGGL(gcnii_prep_kernel, dim3(8, NL), dim3(256), 0, (hipStream_t)stream, W, w_stride, lamda, alpha, VT, V, U);
    ERC_LAUNCH_CHECK("gcnii_chain_prep");
    return ERC_OK;
}

// parts / rows per workgroup for B dialogues of at most T utterances and Mo modalities, and how many dialogues one launch
// may take (all workgroups of a dialogue resident)
extern "C" int erc_gcnii_chain_config(int B, int T, int Mo, int P, int* parts, int* rows, int* dialogues_per_launch) {
    ERC_REQUIRE(B > 0 && T > 0 && T <= MAXT && Mo >= 2 && Mo <= 3 && P >= T && parts && rows && dialogues_per_launch,
                "gcnii_chain_config: B=%d T=%d modalities=%d (T <= %d)", B, T, Mo, MAXT);
    int dev = 0, nf = 0, nb = 0;
    hipDeviceProp_t prop;
    ERC_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess, "gcnii_chain_config: no device");
    const int lds = chain_lds(P);
    ERC_REQUIRE(lds <= 160 * 1024, "gcnii_chain_config: %d bytes of LDS", lds);
    ERC_REQUIRE(chain_ensure_lds(gcnii_chain_kernel<false>, lds) && chain_ensure_lds(gcnii_chain_kernel<true>, lds), "gcnii_chain_config: LDS limit");
    ERC_REQUIRE(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, gcnii_chain_kernel<false>, CNT, lds) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gcnii_chain_kernel<true>, CNT, lds) == hipSuccess && nf >= 1 && nb >= 1,
                "gcnii_chain_config: occupancy query");
    const int cap = prop.multiProcessorCount;              // one workgroup per CU
    int np = (T + MAXRW - 1) / MAXRW;                      // fewest parts (32 rows each)
    const int want = cap / (B * Mo);                       // as many parts as fit with all dialogues in one launch
    if (want > np) np = want;
    if (np > (T + 15) / 16) np = (T + 15) / 16;            // no point below 16 rows per workgroup
    ERC_REQUIRE(Mo * np <= cap, "gcnii_chain_config: a dialogue does not fit the device");
    *parts = np, *rows = (T + np - 1) / np;
    int dpl = cap / (Mo * np);
    *dialogues_per_launch = dpl < B ? dpl : B;
    return ERC_OK;
}

static uint64_t* g_chain_stamps = nullptr;
// diagnostic: phase stamps of workgroup 0, [64 layers][16] (tools/chain_stamps.py); nullptr switches them off
extern "C" int erc_gcnii_chain_set_stamps(uint64_t* stamps) {
    g_chain_stamps = stamps;
    return ERC_OK;
}
static int chain_launch(bool bwd, const float* ADJ, int P, const float* CR, const int32_t* node_off, int N, int Mo, int B,
                        int parts, int rows, int dpl, const float* W, const float* Call, int ldc, float* HD, int64_t hd_plane,
                        float* ZS, float* DG, float* DZ, int lds_, const float* dHin, float* dHout, float* ZX, int32_t* state,
                        float drop_p, const uint64_t* rng, uint64_t rng_stream0, void* stream) {
    const int lds = chain_lds(P);
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < B; b0 += dpl) {
        const int nb = B - b0 < dpl ? B - b0 : dpl;
        Chain p{ADJ, P, CR, node_off, N, Mo, B, b0, parts, rows, W, Call, ldc, HD, hd_plane, ZS, DG, DZ, lds_, dHin, dHout, ZX,
                state + 1 + B, state + 1, state, drop_p, drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, rng, rng_stream0, nullptr};
        p.stamps = b0 == 0 ? g_chain_stamps : nullptr;
        if (bwd) {
            if (!chain_ensure_lds(gcnii_chain_kernel<true>, lds)) return ERC_E_LAUNCH;
            hipLaunchKernelGGL(gcnii_chain_kernel<true>, dim3(nb * Mo * parts), dim3(CNT), lds, st, p);
        } else {
            if (!chain_ensure_lds(gcnii_chain_kernel<false>, lds)) return ERC_E_LAUNCH;
            hipLaunchKernelGGL(gcnii_chain_kernel<false>, dim3(nb * Mo * parts), dim3(CNT), lds, st, p);
        }
        ERC_LAUNCH_CHECK("gcnii_chain");
    }
    return ERC_OK;
}

extern "C" int erc_gcnii_chain_fwd(const float* ADJ, int P, const float* CR, const int32_t* node_off, int N, int Mo, int B, int T,
                                   int parts, int rows, int dialogues_per_launch, const float* VT, const float* Call, int ldc,
                                   float* HD, int64_t hd_plane, float* ZS, int lds, float* ZX, int32_t* state, float drop_p,
                                   const uint64_t* rng_state, uint64_t rng_stream0, void* stream) {
    ERC_REQUIRE(ADJ && CR && node_off && VT && Call && HD && ZS && ZX && state, "gcnii_chain_fwd: null pointer");
    ERC_REQUIRE(N > 0 && Mo >= 2 && Mo <= 3 && B > 0 && T > 0 && T <= MAXT && P >= T && parts >= 1 && rows >= 1 && rows <= MAXRW &&
                    parts * rows >= T && dialogues_per_launch >= 1 && ldc >= NL * FD && lds >= NL * FD,
                "gcnii_chain_fwd: bad sizes (T=%d parts=%d rows=%d)", T, parts, rows);
    ERC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng_state), "gcnii_chain_fwd: drop_p=%f", (double)drop_p);
    return chain_launch(false, ADJ, P, CR, node_off, N, Mo, B, parts, rows, dialogues_per_launch, VT, Call, ldc, HD, hd_plane, ZS,
                        nullptr, nullptr, lds, nullptr, nullptr, ZX, state, drop_p, rng_state, rng_stream0, stream);
}

extern "C" int erc_gcnii_chain_bwd(const float* ADJ, int P, const float* CR, const int32_t* node_off, int N, int Mo, int B, int T,
                                   int parts, int rows, int dialogues_per_launch, const float* V, const float* HD,
                                   int64_t hd_plane, const float* dHin, float* dHout, float* DG, float* DZ, int lds, float* ZX,
                                   int32_t* state, float drop_p, void* stream) {
    ERC_REQUIRE(ADJ && CR && node_off && V && HD && dHin && dHout && DG && DZ && ZX && state, "gcnii_chain_bwd: null pointer");
    ERC_REQUIRE(N > 0 && Mo >= 2 && Mo <= 3 && B > 0 && T > 0 && T <= MAXT && P >= T && parts >= 1 && rows >= 1 && rows <= MAXRW &&
                    parts * rows >= T && dialogues_per_launch >= 1 && lds >= NL * FD,
                "gcnii_chain_bwd: bad sizes (T=%d parts=%d rows=%d)", T, parts, rows);
    return chain_launch(true, ADJ, P, CR, node_off, N, Mo, B, parts, rows, dialogues_per_launch, V, nullptr, 0,
                        const_cast<float*>(HD), hd_plane, nullptr, DG, DZ, lds, dHin, dHout, ZX, state, drop_p, nullptr, 0, stream);
}
