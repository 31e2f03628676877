// Batched weight gradients of the COGMEN bf16 compute mode: every dW = A^T B product of the step (A [K, M <= 128] and
// B [K, N] both K-major and bf16 in memory, K = #nodes) as ONE launch on v_mfma_f32_16x16x32_bf16, fp32 accumulation.
//
// Replaces the autograd weight-gradient GEMMs behind loss.backward() (reference: track_mm/cogmen.py:187-189) in that
// mode; csrc/wgrad.hip stays the kernel of the fp32 parity path and of the other modules.
//
// Why a second kernel (measured on wgrad.hip: 20.4 us at N = 1982, 171 us at N = 33 k, 43 MB of HBM traffic for 17.8 MB of
// operands and 1.1 MB of gradients):
//  * the gradient operands were fp32 in memory and rounded to bf16 in registers.  The backward tile kernel and the head now
//    STORE them as bf16 (the rounding moves from the load to the store: the same products), which halves their bytes -- and
//    bytes through the per-CU L2 path, not MFMA time, set the K loop (findings 5 / 33 of DESIGN.md);
//  * wave tile 128 x 64 instead of 64 x 64: every product of this mode has one 100-wide operand (dH0, dH1, H1, dZ, Z), so
//    M <= 128 covers it with ONE row of tiles and the wide operand (X 1380, M 900, dQKVS 400 columns) is read exactly once;
//    a 16-byte load gives a lane 8 neighbouring output rows for one k, an 8-byte load 4 neighbouring columns:
//    384 bytes of operands per k-row feed 128 x 64 outputs (42 FLOP/B; 64 x 64 with an fp32 operand: 21);
//  * 47 tiles instead of 94 and 5 instead of 8 splits at N = 1982: 235 work items, all co-resident, 7 MB of slabs
//    instead of 12; the 128 accumulator registers go to the AGPR half of the unified register file (4 wavefronts per
//    workgroup, one workgroup per CU), which leaves the VGPRs for two 8-step load groups in flight per wavefront.
//
// Operand access is the K-permutation trick of wgrad.hip: the matrix core sums over all 32 (lane group, slot) pairs
// whichever k sits in which slot, so a lane fills its 8 slots from 8 consecutive k-steps of its wavefront (slot j = k-step
// j's k = 4 ks + g, for A and B alike) and no operand is transposed: fragment i of A (output rows m0 + 8 r + i) takes
// element i of each of the 8 loads -- one v_perm_b32 per pair of slots.
//
// K is split over the 4 wavefronts of a workgroup (reduced through LDS) and over `splits` workgroups per tile (partial
// tiles through a slab with write-through stores + an arrival counter; the last arriver adds them in split order): no
// float atomics, bit-reproducible.
#include "erc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define ERC_GLOBAL __attribute__((address_space(1)))

constexpr int W2_IDX_CAP = 4096;            // k per work item (row-gather stage in LDS)
constexpr int W2_SLAB = 128 * 64 + 192;     // floats per partial tile: 128 x 64 + the two bias strips
constexpr int W2_MAX_DESC = 16;

struct W2Desc {  // 112 bytes; mirrored by engine.GemmPlanner.flush_wgrads_bf16 ("<QQQQQQQ14i")
    const unsigned short* A;  // bf16 [K, lda]: M <= 128 columns used; lda % 8 == 0, lda >= 8 ceil(M / 8), pad columns finite
    const unsigned short* B;  // bf16 [K or gathered rows, ldb]: ldb % 4 == 0, ldb >= 4 ceil(N / 4), pad columns finite
    float* C;                 // ct == 0: C[m * ldc + n]; ct == 1: C[n * ldc + m]
    float* bias_a;            // [M] column sums of A over k (or null)
    float* bias_b;            // [N] column sums of B over k (or null)
    const int32_t* b_gather;  // row of B for every k, or null
    const int32_t* k_dev;     // capacity mode: the true K lives on the device (K below = the capacity the splits are cut for), or null
    int lda, ldb, ldc, M, N, K;
    int ct, cvec, splits, tiles_n, item_base, n_items, tile_base;   // cvec: 16-byte stores of C are legal
    int pad0;
};
static_assert(sizeof(W2Desc) == 112, "W2Desc layout");

__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte write-through store / cache-bypassing load of a slab quad (sc0 sc1 = system scope: the partial tiles cross XCDs,
// whose L2s are not coherent with each other).  The compiler does not count inline-asm loads: the caller waits (vmcnt).
__device__ __forceinline__ void st_sc1_x4(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 ld_sc1_x4(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float bf_lo(unsigned d) { return __builtin_bit_cast(float, d << 16); }
__device__ __forceinline__ float bf_hi(unsigned d) { return __builtin_bit_cast(float, d & 0xffff0000u); }
// {lo half of x, lo half of y} / {hi half of x, hi half of y} as one dword (x in the low 16 bits)
__device__ __forceinline__ unsigned perm_lo(unsigned x, unsigned y) { return __builtin_amdgcn_perm(y, x, 0x05040100u); }
__device__ __forceinline__ unsigned perm_hi(unsigned x, unsigned y) { return __builtin_amdgcn_perm(y, x, 0x07060302u); }

// diagnostic phase stamps (tools/wgrad_stamps.py): the 100 MHz real-time counter, thread 0 of one work item (slots 0..7) and
// of the last arriver of that item's tile (slots 8..11)
#define W2_STAMP(slot)                                                                          \
    do {                                                                                        \
        if (stamps && threadIdx.x == 0) stamps[slot] = __builtin_amdgcn_s_memrealtime();        \
    } while (0)

__device__ __forceinline__ void w2_body(const W2Desc& d, const int local, float* red, float* bred, int* idx, int* s_flag,
                                        float* slabs, int* counters, uint64_t* stamps_item, uint64_t* stamps_tile) {
    uint64_t* stamps = stamps_item;
    W2_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int split = local % d.splits, tn = local / d.splits;
    const int n0 = tn * 64;
    const int nks = (d.K + 3) >> 2;
    const int per = (nks + d.splits - 1) / d.splits;
    const int ks_begin = split * per, ks_end = min(nks, ks_begin + per);
    const int k_begin = ks_begin * 4;
    const int nk = max(1, min(d.K, ks_end * 4) - k_begin);
    // capacity mode: rows [K_true, K) hold stale values of earlier steps -- they are read (in bounds) and masked
    const int K_true = d.k_dev ? min(*(const ERC_GLOBAL int32_t*)d.k_dev, d.K) : d.K;
    const ERC_GLOBAL int32_t* const gather = (const ERC_GLOBAL int32_t*)d.b_gather;
    const bool gath = gather != nullptr;   // uniform
    const ERC_GLOBAL unsigned short* const Ag = (const ERC_GLOBAL unsigned short*)d.A;
    const ERC_GLOBAL unsigned short* const Bg = (const ERC_GLOBAL unsigned short*)d.B;

    // column offsets of this lane (clamped: out-of-range columns only feed output rows / columns that are never stored)
    const int ma = 8 * r, nb = n0 + 4 * r;
    const int a_c = ma < d.M ? ma : 0;
    const int b_c = nb < d.N ? nb : 0;
    const int ns = (max(0, ks_end - ks_begin) + 3) >> 2;      // k-steps of this wavefront: ks_begin + w + 4 s, s < ns

    // Every global load is unconditional (a guarded load is a branch + a full wait, finding 1): an out-of-range k reads a
    // clamped row and its words are ANDed with 0.  A does not depend on the row gather: its first group is requested in
    // front of the gather stage.
    auto load_a = [&](const int s0, u32x4 (&a)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 4 * (ks_begin + w + 4 * (s0 + u)) + g;
            const int kl = max(0, min(k - k_begin, nk - 1));
            a[u] = *(const ERC_GLOBAL u32x4*)(Ag + (int64_t)(k_begin + kl) * d.lda + a_c);
        }
    };
    auto load_b = [&](const int s0, u32x2 (&b)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 4 * (ks_begin + w + 4 * (s0 + u)) + g;
            const int kl = max(0, min(k - k_begin, nk - 1));
            const int64_t brow = (int64_t)(gath ? idx[kl] : k_begin + kl) * d.ldb;
            b[u] = *(const ERC_GLOBAL u32x2*)(Bg + brow + b_c);
        }
    };
    // FOUR 8-step groups in flight per wavefront (192 VGPRs of operand words; the accumulators live in AGPRs): at N = 1982 a
    // wavefront's whole K range (31 steps) is requested before the first product -- one memory latency instead of two
    u32x4 ga0[8], ga1[8], ga2[8], ga3[8];
    u32x2 gb0[8], gb1[8], gb2[8], gb3[8];
    load_a(0, ga0);
    if (gath) {
        for (int t = tid; t < nk; t += 256) idx[t] = gather[k_begin + t];
        __syncthreads();
    }
    load_b(0, gb0);
    load_a(8, ga1), load_b(8, gb1);
    load_a(16, ga2), load_b(16, gb2);
    load_a(24, ga3), load_b(24, gb3);
    W2_STAMP(1);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias strips on the matrix cores too: A^T 1 and 1^T B with an all-ones fragment (bf16 1.0 = 0x3F80) -- 8 + 4 products
    // per group instead of 96 conversions + 96 adds on the VALU (which, not the matrix cores, bounded the K loop)
    f32x4 bacc_a[8], bacc_b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) bacc_a[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) bacc_b[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
    const bool want_a = d.bias_a != nullptr && tn == 0, want_b = d.bias_b != nullptr;   // uniform

    auto mma_group = [&](const int s0, u32x4 (&a)[8], u32x2 (&b)[8]) {
        // k mask of every step, applied to the raw words -- only a group that reaches past the item's k range needs it (uniform)
        const bool ragged = 4 * (ks_begin + 4 * (s0 + 7) + 3) + 3 >= min(K_true, 4 * ks_end);
        if (ragged) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ks = ks_begin + w + 4 * (s0 + u);
                const unsigned km = (ks < ks_end && 4 * ks + g < K_true) ? 0xffffffffu : 0u;
                a[u] = (u32x4){a[u].x & km, a[u].y & km, a[u].z & km, a[u].w & km};
                b[u] = (u32x2){b[u].x & km, b[u].y & km};
            }
        }
        // fragment j of B: slot pair (2 dd, 2 dd + 1) = element j of loads 2 dd, 2 dd + 1
        u32x4 fb[4];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            fb[0][dd] = perm_lo(b[2 * dd].x, b[2 * dd + 1].x), fb[1][dd] = perm_hi(b[2 * dd].x, b[2 * dd + 1].x);
            fb[2][dd] = perm_lo(b[2 * dd].y, b[2 * dd + 1].y), fb[3][dd] = perm_hi(b[2 * dd].y, b[2 * dd + 1].y);
        }
        if (want_b) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bacc_b[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, fb[j]), bacc_b[j], 0, 0, 0);
        }
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {   // fragments 2 ip, 2 ip + 1 of A come from dword ip of the loads
            u32x4 fa0, fa1;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                fa0[dd] = perm_lo(a[2 * dd][ip], a[2 * dd + 1][ip]);
                fa1[dd] = perm_hi(a[2 * dd][ip], a[2 * dd + 1][ip]);
            }
            if (want_a) {
                bacc_a[2 * ip] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa0), ones, bacc_a[2 * ip], 0, 0, 0);
                bacc_a[2 * ip + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa1), ones, bacc_a[2 * ip + 1], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * ip][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa0), __builtin_bit_cast(bf16x8, fb[j]),
                                                                         acc[2 * ip][j], 0, 0, 0);
                acc[2 * ip + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa1), __builtin_bit_cast(bf16x8, fb[j]),
                                                                             acc[2 * ip + 1][j], 0, 0, 0);
            }
        }
    };
    // ring of four groups: a group's buffers are refilled (unconditionally: past the end the clamped rows are re-read and
    // never multiplied) as soon as its products are issued
    for (int s0 = 0; s0 < ns; s0 += 32) {
        __builtin_amdgcn_sched_barrier(0);
        mma_group(s0, ga0, gb0);
        __builtin_amdgcn_sched_barrier(0);
        load_a(s0 + 32, ga0), load_b(s0 + 32, gb0);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + 8 < ns) mma_group(s0 + 8, ga1, gb1);     // (uniform)
        __builtin_amdgcn_sched_barrier(0);
        load_a(s0 + 40, ga1), load_b(s0 + 40, gb1);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + 16 < ns) mma_group(s0 + 16, ga2, gb2);
        __builtin_amdgcn_sched_barrier(0);
        load_a(s0 + 48, ga2), load_b(s0 + 48, gb2);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + 24 < ns) mma_group(s0 + 24, ga3, gb3);
        __builtin_amdgcn_sched_barrier(0);
        load_a(s0 + 56, ga3), load_b(s0 + 56, gb3);
    }
    W2_STAMP(2);
    // ---- bias strips of this wavefront -> LDS (summed over the wavefronts below).  bacc_a[i][q] of lane ln = sum_k A[k][m = 8 (4 (ln >> 4)
    //      + q) + i], the same in all 16 columns: lanes with (ln & 15) == 0 write; bacc_b[j][q] = sum_k B[k][n0 + 4 (ln & 15) + j],
    //      the same in all rows: lanes < 16 write row q = 0
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) bred[w * 192 + 8 * (4 * g + q) + i] = bacc_a[i][q];
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bred[w * 192 + 128 + 4 * lane + j] = bacc_b[j][0];
    }

    // ---- epilogue.  acc[i][j][q] of lane ln is output (m = 8 (4 (ln >> 4) + q) + i, n = n0 + 4 (ln & 15) + j).  Four
    //      quarters h (two per pass through LDS, 16 KB per wavefront and pass), a float4 per (lane, register quad):
    //        ct == 0: pass h = rows i in {2 h, 2 h + 1}, float4 over j       -> 16 bytes along n of C[m][n]
    //        ct == 1: pass h = column j = h, float4 over i in {4 a .. 4 a + 3} -> 16 bytes along m of C[n][m]
    float* const slab = slabs + (int64_t)(d.item_base + local) * W2_SLAB;
    const bool direct = d.splits == 1;
    const bool ct = d.ct != 0;
    auto store_c = [&](const int f4, const int h, const f32x4 v) {
        const int il = f4 >> 8, q = (f4 >> 6) & 3, ln = f4 & 63;
        const int rp = 4 * (ln >> 4) + q;
        ERC_GLOBAL float* const Cg = (ERC_GLOBAL float*)d.C;
        if (!ct) {
            const int m = 8 * rp + 2 * h + il, n = n0 + 4 * (ln & 15);
            if (m >= d.M || n >= d.N) return;
            ERC_GLOBAL float* dst = Cg + (int64_t)m * d.ldc + n;
            if (d.cvec && n + 3 < d.N) {
                *(ERC_GLOBAL f32x4*)dst = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < d.N) dst[j] = v[j];
            }
        } else {
            const int m = 8 * rp + 4 * il, n = n0 + 4 * (ln & 15) + h;
            if (m >= d.M || n >= d.N) return;
            ERC_GLOBAL float* dst = Cg + (int64_t)n * d.ldc + m;
            if (d.cvec && m + 3 < d.M) {
                *(ERC_GLOBAL f32x4*)dst = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (m + i < d.M) dst[i] = v[i];
            }
        }
    };
#pragma unroll
    for (int P = 0; P < 2; ++P) {      // two passes of 16 KB per wavefront: h = 2 P, 2 P + 1
        __syncthreads();   // pass 0: idx[] / bred complete; pass 1: red reuse
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int h = 2 * P + hh;
                    f32x4 v;
                    if (!ct) v = (f32x4){acc[2 * h + il][0][q], acc[2 * h + il][1][q], acc[2 * h + il][2][q], acc[2 * h + il][3][q]};
                    else v = (f32x4){acc[4 * il][h][q], acc[4 * il + 1][h][q], acc[4 * il + 2][h][q], acc[4 * il + 3][h][q]};
                    *reinterpret_cast<f32x4*>(red + w * 4096 + (((hh * 2 + il) * 4 + q) * 64 + lane) * 4) = v;
                }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f4 = tid + 256 * u;      // quad (hh = u >> 1, il, q, lane) of this pass
            f32x4 s = *reinterpret_cast<const f32x4*>(red + f4 * 4);
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(red + ww * 4096 + f4 * 4);
                s += t;
            }
            if (direct) {
                store_c(tid + 256 * (u & 1), 2 * P + (u >> 1), s);
            } else {  // slab layout [h][u & 1][thread] quads: one 16-byte write-through store per thread, 1 KB runs per wave instruction
                st_sc1_x4(slab + ((4 * P + u) * 256 + tid) * 4, s);
            }
        }
    }
    if (tid < 192) {
        const float v = ((bred[tid] + bred[192 + tid]) + bred[384 + tid]) + bred[576 + tid];
        if (direct) {
            if (tid < 128) {
                if (want_a && tid < d.M) ((ERC_GLOBAL float*)d.bias_a)[tid] = v;
            } else if (want_b && n0 + tid - 128 < d.N) {
                ((ERC_GLOBAL float*)d.bias_b)[n0 + tid - 128] = v;
            }
        } else {
            st_sc1(slab + 8192 + tid, v);
        }
    }
    W2_STAMP(3);
    if (direct) return;

    // ---- publish the partial tile; the workgroup that arrives last adds the slabs in split order
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    W2_STAMP(4);
    int* const counter = counters + d.tile_base + tn;
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == d.splits - 1;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        *s_flag = last;
    }
    __syncthreads();
    W2_STAMP(5);
    if (!*s_flag) return;
    stamps = stamps_tile;
    W2_STAMP(8);
    const float* const tile_slabs = slabs + (int64_t)(d.item_base + tn * d.splits) * W2_SLAB;
    // the 8 quads of a thread, 4 splits per batch: 32 independent 16-byte loads in flight (the accumulators are dead by now).
    // The wait is an asm statement that takes the results as in/out operands, so nothing that uses them can be scheduled in
    // front of it (an asm statement takes at most 30 operands: two statements of 16).
    f32x4 sum[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) sum[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int sp0 = 0; sp0 < d.splits; sp0 += 4) {
        f32x4 v[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* p = tile_slabs + (int64_t)min(sp0 + j, d.splits - 1) * W2_SLAB + tid * 4;
#pragma unroll
            for (int x = 0; x < 8; ++x) v[j][x] = ld_sc1_x4(p + x * 1024);
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj += 2)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(v[jj][0]), "+v"(v[jj][1]), "+v"(v[jj][2]), "+v"(v[jj][3]), "+v"(v[jj][4]), "+v"(v[jj][5]), "+v"(v[jj][6]),
                           "+v"(v[jj][7]), "+v"(v[jj + 1][0]), "+v"(v[jj + 1][1]), "+v"(v[jj + 1][2]), "+v"(v[jj + 1][3]),
                           "+v"(v[jj + 1][4]), "+v"(v[jj + 1][5]), "+v"(v[jj + 1][6]), "+v"(v[jj + 1][7])
                         :
                         : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mk = sp0 + j < d.splits ? 1.f : 0.f;
#pragma unroll
            for (int x = 0; x < 8; ++x) sum[x] += v[j][x] * mk;
        }
    }
    W2_STAMP(9);
#pragma unroll
    for (int x = 0; x < 8; ++x) store_c(tid + 256 * (x & 1), x >> 1, sum[x]);
    if (tid < 192) {
        float v = 0.f;
        for (int sp0 = 0; sp0 < d.splits; sp0 += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = ld_sc1(tile_slabs + (int64_t)min(sp0 + j, d.splits - 1) * W2_SLAB + 8192 + tid);
#pragma unroll
            for (int j = 0; j < 8; ++j) v += t[j] * (sp0 + j < d.splits ? 1.f : 0.f);
        }
        if (tid < 128) {
            if (want_a && tid < d.M) ((ERC_GLOBAL float*)d.bias_a)[tid] = v;
        } else if (want_b && n0 + tid - 128 < d.N) {
            ((ERC_GLOBAL float*)d.bias_b)[n0 + tid - 128] = v;
        }
    }
    W2_STAMP(10);
}

struct W2Bases {  // first work item of every descriptor, passed by value (no dependent table reads to find one's descriptor)
    int v[W2_MAX_DESC];
};

__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const W2Desc* __restrict__ table, const int n_desc, const W2Bases bases,
                                                         float* slabs, int* counters, uint64_t* stamps, int stamp_item) {
    __shared__ __attribute__((aligned(16))) float red[4 * 4096];   // 64 KB: two reduction passes
    __shared__ float bred[4 * 192];
    __shared__ int idx[W2_IDX_CAP];
    __shared__ int s_flag;
    const int L = blockIdx.x;
    int di = 0;
#pragma unroll
    for (int t = 1; t < W2_MAX_DESC; ++t)
        if (t < n_desc && L >= bases.v[t]) di = t;
    const W2Desc d = table[di];
    const int local = L - d.item_base;
    if (local >= d.n_items) return;
    // (stamp_item is a work item of record 0)
    const bool st_item = stamps && L == stamp_item, st_tile = stamps && di == 0 && local / d.splits == stamp_item / d.splits;
    w2_body(d, local, red, bred, idx, &s_flag, slabs, counters, st_item ? stamps : nullptr, st_tile ? stamps : nullptr);
}

}  // namespace

static uint64_t* g_w2_stamps = nullptr;
static int g_w2_stamp_item = 0;
// diagnostic: 16 x uint64 phase stamps (10 ns ticks) of work item `item` (a work item of record 0) of the following launches
extern "C" int erc_wgrad_bf16_set_stamps(uint64_t* stamps, int item) {
    g_w2_stamps = stamps, g_w2_stamp_item = item;
    return ERC_OK;
}
extern "C" int64_t erc_wgrad_bf16_slab_floats(void) { return W2_SLAB; }
extern "C" int erc_wgrad_bf16_max_k_per_split(void) { return W2_IDX_CAP; }

// table: n_desc W2Desc records (device memory, <= 16); item_base: HOST array of the records' item_base fields; n_items = sum
// of tiles * splits; slabs: n_items * erc_wgrad_bf16_slab_floats() floats; counters: one zero-initialised int32 per output
// tile (left zero by the launch).
extern "C" int erc_wgrad_bf16(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                              int32_t* counters, void* stream) {
    ERC_REQUIRE(table && item_base && n_desc > 0 && n_desc <= W2_MAX_DESC && n_items > 0 && slabs && counters,
                "wgrad_bf16: bad arguments (at most %d records per launch)", W2_MAX_DESC);
    W2Bases bases{};
    for (int t = 0; t < n_desc; ++t) {
        ERC_REQUIRE(item_base[t] >= 0 && item_base[t] < n_items && (t == 0 ? item_base[0] == 0 : item_base[t] > item_base[t - 1]),
                    "wgrad_bf16: item_base[%d] = %d", t, item_base[t]);
        bases.v[t] = item_base[t];
    }
    hipLaunchKernelGGL(wgrad_bf16_kernel, dim3(n_items), dim3(256), 0, (hipStream_t)stream, (const W2Desc*)table, n_desc, bases,
                       slabs, counters, g_w2_stamps, g_w2_stamp_item);
    ERC_LAUNCH_CHECK("wgrad_bf16");
    return ERC_OK;
}
