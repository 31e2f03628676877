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
#include "optim_dev.h"
#include "split_dev.h"
#include <string.h>

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
    int kind;                 // 0: product record; 1 (fused optimizer only): plain range -- C[0, M) are finished gradients written
                              // by an earlier launch (BatchNorm's scale / shift): one work item applies the update to them
};

// THE OPTIMIZER INSIDE THIS LAUNCH (wgrad_bf16_kernel<true>; single-rank steps).  A first version let the LAST ARRIVER of a
// tile apply Adam to the whole 128 x 64 tile it had just summed: 48 us -- the update of 280 k parameters (index decomposition
// for five bf16 shadow layouts per quad) ran on the 47 CUs that happened to arrive last instead of on 274 workgroups.  Here
// the final reduction is a REDUCE-SCATTER: every split of a tile publishes its partial tile, waits until all S splits of
// the tile have (monotonic per-tile counter against S x the launch sequence number; bounded; every item of the launch is
// resident: the host admits <= 256), and then finishes the quads x with x % S == its split index -- sums them over the S
// slabs in split order, writes the gradient and applies torch.optim.Adam / AdamW to the same elements of the flat
// parameter / moment buffers (same layout as the gradient), bf16 shadows included.  The update is spread over all work items,
// the slab sums over S times as many threads, and the optimizer launch (9.8 us + a launch gap of a 96 us step) is gone.
// Every workgroup keeps private copies of the step count (state[4 + b], as adam_kernel) and of the launch sequence number.
struct W2Adam {
    int decoupled, spin_limit;     // spin_limit: bound of the wait for a tile's splits (test hook: erc_wgrad_bf16_set_spin_limit)
    float lr, b1, b2, eps, wd, grad_scale;
    float *data, *grad, *m, *v;
    int64_t* state;
    unsigned short* shadow;
    const int32_t* skip;
    int32_t* seq;          // [512] private launch sequence numbers
    int32_t* health;       // raised when the wait for a tile's splits times out
    ShadowTab tab;
    // data parallel, ERC_DP_P2P=1 (engine.P2PExchange; csrc/optim.hip P2PArgs): the gradient exchange INSIDE this launch -- a work
    // item that has summed its quads over the tile's splits publishes them (write-through, system scope) at their flat gradient
    // offsets in this rank's publish buffer, posts (epoch, health bit) to every rank's flag array, waits (bounded) for the same
    // item of every rank and continues with the RANK-ORDERED sum: bit-identical replicas, no RCCL call, no optimizer launch.
    int x_world, x_rank, x_spin, x_pad;
    float* x_pub[8];
    int32_t* x_flags[8];
    int64_t* x_epoch;      // [512] one exchange counter per work item (private to its workgroup)
    int64_t x_npad;
};
static_assert(sizeof(W2Desc) == 112, "W2Desc layout");

__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte write-through store / cache-bypassing load of a slab quad (sc0 sc1 = system scope: the partial tiles cross XCDs,
// whose L2s are not coherent with each other).  The compiler does not count inline-asm loads: the caller waits (vmcnt).
__device__ __forceinline__ void st_sc1_x4(float* p, f32x4 v) {
    // (s_nop: a store of more than 64 bits needs wait states before its data registers may be overwritten, and the compiler's
    //  hazard recognizer does not look inside an asm statement -- without them a v_cndmask scheduled right behind the store
    //  replaced the last dword of lanes 12-15 of every 16: finding 44)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
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

// WIDE (large K: B = 512 batches): the four wavefronts of a workgroup do not split K but take four NEIGHBOURING column tiles
// of the record over the same k-steps -- they request the same A rows at about the same time, so the 256 B of an A row per
// k reach the CU once instead of once per column tile (at N = 33 k the A operand, 6.6 MB per record, does not fit a 4 MB L2 and
// was streamed through the fabric 22 times for the 22 column tiles of the projection gradient: 397 of the launch's 595 MB).
template <bool ADAM, bool WIDE, int NT>
__device__ __forceinline__ void w2_body(const W2Desc& d, const int local, float* red, float* bred, int* idx, int* s_flag,
                                        float* slabs, int* counters, uint64_t* stamps_item, uint64_t* stamps_tile, const W2Adam& ad) {
    uint64_t* stamps = stamps_item;
    W2_STAMP(0);
    // fused optimizer: the step's skip decision and this workgroup's private step count, requested before the K loop
    bool adam_on = false;
    int64_t adam_step = 0;
    AdamCoef ac;
    if (ADAM && d.kind == 1) {
        const bool skip = ad.skip && __hip_atomic_load(ad.skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        adam_step = ad.state[4 + blockIdx.x] + 1;
        adam_on = !skip;
        ac.init(ad.lr, ad.b1, ad.b2, ad.eps, ad.wd, ad.decoupled, ad.grad_scale, adam_step);
    }
    // (the counts are WRITTEN behind a workgroup barrier: a wavefront that starts late must not read a bumped value)
    auto bump_step = [&]() __attribute__((always_inline)) {
        if (!ADAM || !adam_on) return;
        if (threadIdx.x == 0) {
            ad.state[4 + blockIdx.x] = adam_step;
            if (blockIdx.x == 0) ad.state[0] = adam_step, ad.state[1] += 1;     // nobody reads them during this launch
        }
        if (blockIdx.x == 0)                                                    // the copies no workgroup of this launch owns
            for (int t = gridDim.x + threadIdx.x; t < 512; t += 256) ad.state[4 + t] = adam_step;
    };
    // one finished gradient quad at flat offset `off`: the update of the same four elements of data / m / v + their shadows
    // the shadow table is read from an LDS copy (the reduction buffer is free by then): as a kernel argument its 130 words
    // were held in SGPRs across the whole kernel (~700 spilled)
    const ShadowTab& stab = *reinterpret_cast<const ShadowTab*>(red);
    auto copy_tab = [&]() __attribute__((always_inline)) {
        ShadowTab* const wt = reinterpret_cast<ShadowTab*>(red);
        wt->n = ad.tab.n, wt->flags = ad.tab.flags;
#pragma unroll
        for (int t = 0; t < SHADOW_MAX; ++t) {      // (field by field: a struct assignment goes through a private copy)
            const ShadowDesc& a = ad.tab.d[t];
            ShadowDesc& b = wt->d[t];
            b.src_off = a.src_off, b.n_el = a.n_el, b.dst_off = a.dst_off, b.n0 = a.n0, b.n1 = a.n1, b.sn0 = a.sn0, b.sn1 = a.sn1;
            b.sn2 = a.sn2, b.sk0 = a.sk0, b.sk1 = a.sk1, b.sk2 = a.sk2, b.ld = a.ld, b.mode = a.mode;
            b.plane_stride = a.plane_stride, b.terms = a.terms;
        }
    };
    auto adam_quad = [&](const int64_t off, const f32x4 gq, const f32x4 pq, const f32x4 mq, const f32x4 vq) __attribute__((always_inline)) -> float4 {
        float4 pv = make_float4(pq.x, pq.y, pq.z, pq.w), mv = make_float4(mq.x, mq.y, mq.z, mq.w), vv = make_float4(vq.x, vq.y, vq.z, vq.w);
        ac.upd(pv.x, gq.x, mv.x, vv.x), ac.upd(pv.y, gq.y, mv.y, vv.y), ac.upd(pv.z, gq.z, mv.z, vv.z), ac.upd(pv.w, gq.w, mv.w, vv.w);
        *(ERC_GLOBAL f32x4*)(ad.data + off) = (f32x4){pv.x, pv.y, pv.z, pv.w};
        *(ERC_GLOBAL f32x4*)(ad.m + off) = (f32x4){mv.x, mv.y, mv.z, mv.w}, *(ERC_GLOBAL f32x4*)(ad.v + off) = (f32x4){vv.x, vv.y, vv.z, vv.w};
        return pv;
    };
    // the bf16 shadows of up to 8 updated quads: ONE pass over the table's descriptors (read from the LDS copy, not unrolled)
    auto shadow_quads = [&](const int64_t (&offs)[8], const float4 (&pn)[8], const int count, const bool (&isq)[8], const unsigned tmask) __attribute__((always_inline)) {
        if (!ad.shadow) return;
        const bool quads = (stab.flags & 1) != 0;
        const int nt = stab.n;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            if (o >= count || !isq[o]) continue;
            const int64_t i0 = offs[o];
            const float4 pv = pn[o];
#pragma unroll 1
            for (int t = 0; t < nt; ++t) {
                if (!((tmask >> t) & 1)) continue;      // (uniform) ranges that do not meet this record's gradient
                const ShadowDesc& sd = stab.d[t];
                if (quads) {
                    shadow_store4_desc(ad.shadow, sd, (stab.flags & (2 << t)) != 0, i0, pv);
                } else {
                    shadow_store_desc(ad.shadow, sd, i0, pv.x), shadow_store_desc(ad.shadow, sd, i0 + 1, pv.y);
                    shadow_store_desc(ad.shadow, sd, i0 + 2, pv.z), shadow_store_desc(ad.shadow, sd, i0 + 3, pv.w);
                }
            }
        }
    };
    auto adam_one = [&](const int64_t off, const float gval) __attribute__((always_inline)) {
        float pv = ad.data[off], mv = ad.m[off], vv = ad.v[off];
        ac.upd(pv, gval, mv, vv);
        ad.data[off] = pv, ad.m[off] = mv, ad.v[off] = vv;
        if (ad.shadow) {
            const int nt = stab.n;
#pragma unroll 1
            for (int t = 0; t < nt; ++t) shadow_store_desc(ad.shadow, stab.d[t], off, pv);
        }
    };
    // ---- the exchange of one work item (see W2Adam): `publish` has stored this item's finished gradient elements into the publish
    //      buffer at poff + their flat offsets; returns false when any rank flagged its step invalid (or a peer never arrived)
    __shared__ int s_x_skip;
    auto x_poff = [&]() __attribute__((always_inline)) -> int64_t { return (int64_t)((ad.x_epoch[blockIdx.x] + 1) & 1) * ad.x_npad; };
    auto x_rendezvous = [&](const bool local_skip) __attribute__((always_inline)) -> bool {
        const int b = blockIdx.x, nblk = gridDim.x;
        const int ep = (int)(ad.x_epoch[b] + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's publish stores are out
        __syncthreads();
        if ((int)threadIdx.x < ad.x_world) {
            int32_t *f = nullptr, *mine = nullptr;      // (selected without indexing the by-value struct dynamically)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r == (int)threadIdx.x) f = ad.x_flags[r];
                if (r == ad.x_rank) mine = ad.x_flags[r];
            }
            __hip_atomic_store(f + ad.x_rank * nblk + b, ep * 2 + (local_skip ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const int32_t* w_ = mine + (int)threadIdx.x * nblk + b;
            int val = __hip_atomic_load(w_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), spins = 0;
            while ((val >> 1) - ep < 0) {
                if (++spins > ad.x_spin) {      // a peer never arrived: this ITEM gives up (partial update: fatal for the run, see W2Adam)
                    __hip_atomic_store(ad.health, ERC_HEALTH_RAISED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    val = 2 * ep + 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                val = __hip_atomic_load(w_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const unsigned long long any = __ballot((val & 1) != 0);
            if (threadIdx.x == 0) s_x_skip = any != 0ull;
        }
        __syncthreads();
        if (threadIdx.x == 0) ad.x_epoch[b] = ep;
        return s_x_skip == 0;
    };
    auto x_sum4 = [&](const int64_t poff_off) __attribute__((always_inline)) -> f32x4 {      // rank-ordered sum of one published quad
        f32x4 part[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) part[r] = r < ad.x_world ? ld_sc1_x4(ad.x_pub[r] + poff_off) : (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(part[0]), "+v"(part[1]), "+v"(part[2]), "+v"(part[3]), "+v"(part[4]), "+v"(part[5]), "+v"(part[6]), "+v"(part[7])
                     :
                     : "memory");
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < ad.x_world) acc += part[r];
        return acc;
    };
    auto x_sum1 = [&](const int64_t poff_off) __attribute__((always_inline)) -> float {
        float acc = 0.f;
        for (int r = 0; r < ad.x_world; ++r) {
            const float* pb = nullptr;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q == r) pb = ad.x_pub[q];
            acc += __hip_atomic_load(pb + poff_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return acc;
    };
    if (ADAM && d.kind == 1) {   // plain range of finished gradients: d.C[0, M)
        if (threadIdx.x == 64) copy_tab();
        __syncthreads();
        const int64_t base = d.C - ad.grad;
        int64_t xoff = 0;
        if (ad.x_world > 1) {      // (uniform) exchange the range first: every element published, then summed in rank order
            xoff = x_poff() + base;
            float* mypub = nullptr;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r == ad.x_rank) mypub = ad.x_pub[r];
            for (int e = (int)threadIdx.x; e < d.M; e += 256) __hip_atomic_store(mypub + xoff + e, d.C[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            adam_on = x_rendezvous(!adam_on) && adam_on;
        }
        bump_step();
        if (adam_on) {
            for (int e = 4 * (int)threadIdx.x; e < d.M; e += 1024) {
                if (e + 3 < d.M && ((base + e) & 3) == 0) {
                    int64_t offs[8] = {base + e};
                    float4 pn[8];
                    const bool isq[8] = {true};
                    f32x4 gq = *(const ERC_GLOBAL f32x4*)(d.C + e);
                    if (ad.x_world > 1) {
                        gq = x_sum4(xoff + e);
                        *(ERC_GLOBAL f32x4*)(d.C + e) = gq;
                    }
                    pn[0] = adam_quad(base + e, gq, *(const ERC_GLOBAL f32x4*)(ad.data + base + e),
                                      *(const ERC_GLOBAL f32x4*)(ad.m + base + e), *(const ERC_GLOBAL f32x4*)(ad.v + base + e));
                    shadow_quads(offs, pn, 1, isq, ~0u);
                } else {
                    for (int t = e; t < min(e + 4, d.M); ++t) {
                        float gv = d.C[t];
                        if (ad.x_world > 1) d.C[t] = gv = x_sum1(xoff + t);
                        adam_one(base + t, gv);
                    }
                }
            }
        }
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int split = local % d.splits;
    const int tn = WIDE ? 4 * (local / d.splits) + w : local / d.splits;     // WIDE: the column tile of THIS wavefront (may be >= tiles_n)
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
    // output row of fragment i (0..7) of the A operand for accumulator row group rp (0..15).  bf16 operands: a lane's 16-byte load is
    // 8 neighbouring rows (m = 8 rp + i).  fp32 operands (split modes): TWO 16-byte loads per k -- rows 4 rp .. 4 rp + 3 and
    // 64 + 4 rp .. + 3 -- so that the 16 lanes of a row read 256 contiguous bytes per instruction (8 neighbouring rows per lane as
    // two loads touched every cache line of the row twice: K loop 12.2 us at config 2)
    auto mrow = [](const int rp, const int i) __attribute__((always_inline)) -> int {
        return NT == 1 ? 8 * rp + i : (i < 4 ? 4 * rp + i : 64 + 4 * rp + (i - 4));
    };
    const int ma = NT == 1 ? 8 * r : 4 * r, nb = n0 + 4 * r;
    const int a_c = ma < d.M ? ma : 0;
    const int b_c = nb < d.N ? nb : 0;
    // k-steps of this wavefront: ks_begin + w + 4 s (WIDE: ks_begin + s), s < ns
    const int ns = WIDE ? max(0, ks_end - ks_begin) : (max(0, ks_end - ks_begin) + 3) >> 2;
    auto kstep = [&](const int sidx) __attribute__((always_inline)) { return WIDE ? ks_begin + sidx : ks_begin + w + 4 * sidx; };

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

    if constexpr (NT == 1) {
        // Every global load is unconditional (a guarded load is a branch + a full wait, finding 1): an out-of-range k reads a
        // clamped row and its words are ANDed with 0.  A does not depend on the row gather: its first group is requested in
        // front of the gather stage.
        auto load_a = [&](const int s0, u32x4 (&a)[8]) {
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 4 * kstep(s0 + u) + g;
                const int kl = max(0, min(k - k_begin, nk - 1));
                a[u] = *(const ERC_GLOBAL u32x4*)(Ag + (int64_t)(k_begin + kl) * d.lda + a_c);
            }
        };
        auto load_b = [&](const int s0, u32x2 (&b)[8]) {
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 4 * kstep(s0 + u) + g;
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

        auto mma_group = [&](const int s0, u32x4 (&a)[8], u32x2 (&b)[8]) {
            // k mask of every step, applied to the raw words -- only a group that reaches past the item's k range needs it (uniform)
            const bool ragged = 4 * (WIDE ? ks_begin + s0 + 7 : ks_begin + 4 * (s0 + 7) + 3) + 3 >= min(K_true, 4 * ks_end);
            if (ragged) {
    #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ks = kstep(s0 + u);
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
    } else {
        // ---- split compute modes: A [K, lda] and B [K or gathered rows, ldb] are FP32 in memory (lda, ldb multiples of 4, 16-byte
        //      aligned rows); a k-step is two 16-byte loads of A (output rows m = 8 r .. 8 r + 7) and one of B (columns 4 r ..), every
        //      value is expanded into NT bf16 terms in registers (csrc/split_dev.h) -- slot pair (2 dd, 2 dd + 1) of fragment i is ONE
        //      v_cvt_pk_bf16_f32 of element i of k-steps 2 dd, 2 dd + 1 -- and the NT (NT + 1) / 2 term products run on the bf16 matrix
        //      cores.  
        const ERC_GLOBAL float* const Af = (const ERC_GLOBAL float*)d.A;
        const ERC_GLOBAL float* const Bf = (const ERC_GLOBAL float*)d.B;
        const int a_c1 = 64 + ma < d.M ? 64 + ma : 0;
        // the ring unit is a HALF group (4 k-steps: 48 operand registers); three of them -- two full 8-step groups in flight spill
        // (the expansion needs ~50 registers of its own), one leaves a single memory latency uncovered per group
        struct Half {
            u32x4 a[8], b[4];
        };
        auto load_half = [&](const int s0, Half& h) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 4 * kstep(s0 + u) + g;
                const int kl = max(0, min(k - k_begin, nk - 1));
                const ERC_GLOBAL float* row = Af + (int64_t)(k_begin + kl) * d.lda;
                h.a[2 * u] = *(const ERC_GLOBAL u32x4*)(row + a_c), h.a[2 * u + 1] = *(const ERC_GLOBAL u32x4*)(row + a_c1);
                const int64_t brow = (int64_t)(gath ? idx[kl] : k_begin + kl) * d.ldb;
                h.b[u] = *(const ERC_GLOBAL u32x4*)(Bf + brow + b_c);
            }
        };
        Half h0, h1, h2;
        if (gath) {
            for (int t = tid; t < nk; t += 256) idx[t] = gather[k_begin + t];
            __syncthreads();
        }
        load_half(0, h0), load_half(4, h1), load_half(8, h2);
        W2_STAMP(1);
        auto f32 = [](unsigned w_) __attribute__((always_inline)) { return __builtin_bit_cast(float, w_); };
        // one 8-step group = halves x (k-steps s0 .. s0 + 3: slot pairs 0, 1) and y (s0 + 4 .. s0 + 7: slot pairs 2, 3)
        auto mma_group = [&](const int s0, Half& x, Half& y) {
            const bool ragged = 4 * (ks_begin + 4 * (s0 + 7) + 3) + 3 >= min(K_true, 4 * ks_end);
            if (ragged) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ks = kstep(s0 + u);
                    const unsigned km = (ks < ks_end && 4 * ks + g < K_true) ? 0xffffffffu : 0u;
                    Half& h = u < 4 ? x : y;
                    const int v = u & 3;
                    h.a[2 * v] = (u32x4){h.a[2 * v].x & km, h.a[2 * v].y & km, h.a[2 * v].z & km, h.a[2 * v].w & km};
                    h.a[2 * v + 1] = (u32x4){h.a[2 * v + 1].x & km, h.a[2 * v + 1].y & km, h.a[2 * v + 1].z & km, h.a[2 * v + 1].w & km};
                    h.b[v] = (u32x4){h.b[v].x & km, h.b[v].y & km, h.b[v].z & km, h.b[v].w & km};
                }
            }
            u32x4 fb[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const Half& h = dd < 2 ? x : y;
                    unsigned tt[NT];
                    sp_split2<NT>(f32(h.b[2 * (dd & 1)][j]), f32(h.b[2 * (dd & 1) + 1][j]), tt);
#pragma unroll
                    for (int t = 0; t < NT; ++t) fb[j][t][dd] = tt[t];
                }
            if (want_b) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = NT - 1; t >= 0; --t)
                        bacc_b[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, fb[j][t]), bacc_b[j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {      // fragment i of A: element i & 3 of the load half i >> 2
                u32x4 fa[NT];
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const Half& h = dd < 2 ? x : y;
                    unsigned tt[NT];
                    sp_split2<NT>(f32(h.a[2 * (2 * (dd & 1)) + (i >> 2)][i & 3]), f32(h.a[2 * (2 * (dd & 1) + 1) + (i >> 2)][i & 3]), tt);
#pragma unroll
                    for (int t = 0; t < NT; ++t) fa[t][dd] = tt[t];
                }
                if (want_a) {
#pragma unroll
                    for (int t = NT - 1; t >= 0; --t)
                        bacc_a[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[t]), ones, bacc_a[i], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = sp_mfma<NT>(fa, fb[j], acc[i][j]);
            }
        };
        for (int s0 = 0; s0 < ns; s0 += 24) {
            __builtin_amdgcn_sched_barrier(0);
            mma_group(s0, h0, h1);
            __builtin_amdgcn_sched_barrier(0);
            load_half(s0 + 12, h0), load_half(s0 + 16, h1);
            __builtin_amdgcn_sched_barrier(0);
            if (s0 + 8 < ns) mma_group(s0 + 8, h2, h0);     // (uniform)
            __builtin_amdgcn_sched_barrier(0);
            load_half(s0 + 20, h2), load_half(s0 + 24, h0);
            __builtin_amdgcn_sched_barrier(0);
            if (s0 + 16 < ns) mma_group(s0 + 16, h1, h2);
            __builtin_amdgcn_sched_barrier(0);
            load_half(s0 + 28, h1), load_half(s0 + 32, h2);
        }
    }
    W2_STAMP(2);
    // ---- everything behind the K loop, for ONE column tile: `tn_e` = the tile, `wsel` = 1 for the wavefronts whose accumulators
    //      belong to it (all four when they split K; WIDE: the one wavefront that owns the tile, the others contribute zeros)
    auto finish = [&](const int tn_e, const float wsel) __attribute__((always_inline)) {
        const int tn = tn_e, n0 = 64 * tn_e;
        const bool want_a = d.bias_a != nullptr && tn == 0, want_b = d.bias_b != nullptr;   // uniform
        // ---- bias strips of this wavefront -> LDS (summed over the wavefronts below).  bacc_a[i][q] of lane ln = sum_k A[k][m = 8 (4 (ln >> 4)
        //      + q) + i], the same in all 16 columns: lanes with (ln & 15) == 0 write; bacc_b[j][q] = sum_k B[k][n0 + 4 (ln & 15) + j],
        //      the same in all rows: lanes < 16 write row q = 0
        if ((lane & 15) == 0) {
    #pragma unroll
            for (int i = 0; i < 8; ++i)
    #pragma unroll
                for (int q = 0; q < 4; ++q) bred[w * 192 + mrow(4 * g + q, i)] = bacc_a[i][q] * wsel;
        }
        if (lane < 16) {
    #pragma unroll
            for (int j = 0; j < 4; ++j) bred[w * 192 + 128 + 4 * lane + j] = bacc_b[j][0] * wsel;
        }

        // ---- epilogue.  acc[i][j][q] of lane ln is output (m = 8 (4 (ln >> 4) + q) + i, n = n0 + 4 (ln & 15) + j).  Four
        //      quarters h (two per pass through LDS, 16 KB per wavefront and pass), a float4 per (lane, register quad):
        //        ct == 0: pass h = rows i in {2 h, 2 h + 1}, float4 over j       -> 16 bytes along n of C[m][n]
        //        ct == 1: pass h = column j = h, float4 over i in {4 a .. 4 a + 3} -> 16 bytes along m of C[n][m]
        float* const slab = slabs + (int64_t)(d.item_base + tn * d.splits + split) * W2_SLAB;
        const bool direct = !ADAM && d.splits == 1;      // (the fused-optimizer kernel always goes through the slab)
        const bool ct = d.ct != 0;
        // where quad (f4, h) of the tile goes: offset into C of its first element, and how many of its four elements
        // (consecutive in memory) exist
        auto quad_addr = [&](const int f4, const int h, int& valid) __attribute__((always_inline)) -> int64_t {
            const int il = f4 >> 8, q = (f4 >> 6) & 3, ln = f4 & 63;
            const int rp = 4 * (ln >> 4) + q;
            if (!ct) {
                const int m = mrow(rp, 2 * h + il), n = n0 + 4 * (ln & 15);
                valid = (m < d.M && n < d.N) ? min(4, d.N - n) : 0;
                return (int64_t)m * d.ldc + n;
            }
            const int m = mrow(rp, 4 * il), n = n0 + 4 * (ln & 15) + h;
            valid = (m < d.M && n < d.N) ? min(4, d.M - m) : 0;
            return (int64_t)n * d.ldc + m;
        };
        auto store_c = [&](const int f4, const int h, const f32x4 v) {
            const int il = f4 >> 8, q = (f4 >> 6) & 3, ln = f4 & 63;
            const int rp = 4 * (ln >> 4) + q;
            ERC_GLOBAL float* const Cg = (ERC_GLOBAL float*)d.C;
            if (!ct) {
                const int m = mrow(rp, 2 * h + il), n = n0 + 4 * (ln & 15);
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
                const int m = mrow(rp, 4 * il), n = n0 + 4 * (ln & 15) + h;
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
                        // (both layouts as VALUES and a select: a branch made the compiler choose between pointers to accumulators,
                        //  which parked six of them in scratch once this code sat inside a lambda)
                        const f32x4 v_n = (f32x4){acc[2 * h + il][0][q], acc[2 * h + il][1][q], acc[2 * h + il][2][q], acc[2 * h + il][3][q]};
                        const f32x4 v_t = (f32x4){acc[4 * il][h][q], acc[4 * il + 1][h][q], acc[4 * il + 2][h][q], acc[4 * il + 3][h][q]};
                        const f32x4 v = ct ? v_t : v_n;
                        *reinterpret_cast<f32x4*>(red + w * 4096 + (((hh * 2 + il) * 4 + q) * 64 + lane) * 4) = v * wsel;
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
        if (ADAM) {
            // ---- reduce-scatter among the S splits of the tile (see W2Adam).  S is a power of two <= 8 (host): this split owns
            // the 8 / S quads x = split + S o of every thread, i.e. exactly 8 slab quads to fetch per thread.
            const int S = d.splits, lgS = 31 - __builtin_clz(S), owned = 8 >> lgS;
            const int64_t cbase = d.C - ad.grad;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this split's partial tile is in memory
            __syncthreads();
            W2_STAMP(4);
            int* const counter = counters + d.tile_base + tn;
            int seq = 0;
            if (tid == 0) {
                seq = ad.seq[blockIdx.x] + 1;
                ad.seq[blockIdx.x] = seq;
                __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 64) copy_tab();
            // while the other splits finish: this split's parameter / moment quads (they do not depend on anybody)
            int64_t off[8];
            int valid[8];
            f32x4 pq[8], mq[8], vq[8];
    #pragma unroll
            for (int o = 0; o < 8; ++o) {
                const int x = split + (o << lgS);
                valid[o] = 0;
                off[o] = cbase;
                if (o < owned) {
                    int vl;
                    const int64_t a = cbase + quad_addr(tid + 256 * (x & 1), x >> 1, vl);
                    valid[o] = (vl == 4 && d.cvec) ? 4 : -vl;       // > 0: one aligned quad; < 0: that many single elements
                    if (vl) off[o] = a;
                    const int64_t oc = valid[o] == 4 ? a : cbase;   // (unconditional loads)
                    pq[o] = *(const ERC_GLOBAL f32x4*)(ad.data + oc), mq[o] = *(const ERC_GLOBAL f32x4*)(ad.m + oc), vq[o] = *(const ERC_GLOBAL f32x4*)(ad.v + oc);
                }
            }
            // (split 0: the bias strip element of this thread, its parameter and moments)
            float* bdst = nullptr;
            if (split == 0 && tid < 192) {
                if (tid < 128) {
                    if (want_a && tid < d.M) bdst = d.bias_a + tid;
                } else if (want_b && n0 + tid - 128 < d.N) {
                    bdst = d.bias_b + (n0 + tid - 128);
                }
            }
            const int64_t boff = bdst ? bdst - ad.grad : cbase;
            float bp = ad.data[boff], bm = ad.m[boff], bv = ad.v[boff];
            const bool skip = ad.skip && __hip_atomic_load(ad.skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            adam_step = ad.state[4 + blockIdx.x] + 1;
            adam_on = !skip;
            ac.init(ad.lr, ad.b1, ad.b2, ad.eps, ad.wd, ad.decoupled, ad.grad_scale, adam_step);
            if (tid == 0) {
                int ok = 1, spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - S * seq < 0) {
                    if (++spins >= ad.spin_limit) {      // a split of this tile never arrived: THIS TILE is left un-updated (the other tiles of
                                                         // the launch update, the step count advances): a partial update, fatal for the run --
                                                         // FlatParams.check_health raises at the epoch's end, checkpoint.save refuses
                        __hip_atomic_store(ad.health, ERC_HEALTH_RAISED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                *s_flag = ok;
            }
            __syncthreads();
            W2_STAMP(5);
            if (ad.x_world <= 1) bump_step();      // (every thread read the step count in front of the barriers; with the
            if (!*s_flag) return;                  //  gradient exchange the count moves behind it: a peer may veto the step)
            stamps = stamps_tile ? stamps_tile : stamps;
            W2_STAMP(8);
            const float* const tile_slabs = slabs + (int64_t)(d.item_base + tn * S) * W2_SLAB;
            f32x4 part[8];      // load L: owned quad o = L / S, slab j = L % S
    #pragma unroll
            for (int L = 0; L < 8; ++L) {
                const int x = split + ((L >> lgS) << lgS), jx = L & (S - 1);
                part[L] = ld_sc1_x4(tile_slabs + (int64_t)jx * W2_SLAB + (x * 256 + tid) * 4);
            }
            float bt[8];      // the bias strip's S partial sums (slots >= S re-read the last slab and are masked)
    #pragma unroll
            for (int jj = 0; jj < 8; ++jj) bt[jj] = split == 0 ? ld_sc1(tile_slabs + (int64_t)min(jj, S - 1) * W2_SLAB + 8192 + (tid < 192 ? tid : 0)) : 0.f;
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(part[0]), "+v"(part[1]), "+v"(part[2]), "+v"(part[3]), "+v"(part[4]), "+v"(part[5]), "+v"(part[6]), "+v"(part[7])
                         :
                         : "memory");
            // the sums in split order (the left fold of the last arriver above: bit-identical gradients)
            const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 res[8];
            if (S == 1) {
    #pragma unroll
                for (int o = 0; o < 8; ++o) res[o] = z + part[o];
            } else if (S == 2) {
    #pragma unroll
                for (int o = 0; o < 8; ++o) res[o] = o < 4 ? (z + part[2 * (o & 3)]) + part[2 * (o & 3) + 1] : z;
            } else if (S == 4) {
    #pragma unroll
                for (int o = 0; o < 8; ++o) res[o] = o < 2 ? (((z + part[4 * (o & 1)]) + part[4 * (o & 1) + 1]) + part[4 * (o & 1) + 2]) + part[4 * (o & 1) + 3] : z;
            } else {
                res[0] = (((((((z + part[0]) + part[1]) + part[2]) + part[3]) + part[4]) + part[5]) + part[6]) + part[7];
    #pragma unroll
                for (int o = 1; o < 8; ++o) res[o] = z;
            }
            W2_STAMP(9);
            float bsum = 0.f;      // bias strip element of this thread (split 0 of the tile)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) bsum += bt[jj] * (jj < S ? 1.f : 0.f);
            if (ad.x_world > 1) {      // (uniform) the gradient exchange of this work item: its owned quads + its bias strip elements
                const int64_t poff = x_poff();
                float* mypub = nullptr;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r == ad.x_rank) mypub = ad.x_pub[r];
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o < owned && valid[o] == 4) st_sc1_x4(mypub + poff + off[o], res[o]);
                if (bdst) __hip_atomic_store(mypub + poff + boff, bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                adam_on = x_rendezvous(!adam_on) && adam_on;
                bump_step();
                if (adam_on) {
#pragma unroll
                    for (int o = 0; o < 8; ++o)
                        if (o < owned && valid[o] == 4) res[o] = x_sum4(poff + off[o]);
                    if (bdst) bsum = x_sum1(poff + boff);
                }
            }
            const bool upd = adam_on;
            unsigned tmask = 0;      // shadow ranges that meet this record's gradient [cbase, cbase + rows * ldc)
            {
                const int64_t lo = cbase, hi = cbase + (int64_t)(ct ? d.N : d.M) * d.ldc;
                const int nt = stab.n;
                for (int t = 0; t < nt; ++t)
                    if (stab.d[t].src_off < hi && stab.d[t].src_off + stab.d[t].n_el > lo) tmask |= 1u << t;
            }
            // One copy of the update code, executed `owned` times on element 0 of the register arrays, which are shifted down
            // after every pass (8 unrolled copies with scalar fallbacks were 24 000 instructions of cold code: the instruction
            // fetch, not the arithmetic, set the 4.3 us this phase took).  Quads are all-or-nothing here: the host only fuses
            // records whose rows are multiples of 4 elements and 16-byte aligned.
    #pragma unroll 1
            for (int it = 0; it < owned; ++it) {
                if (valid[0] == 4) {
                    *(ERC_GLOBAL f32x4*)(ad.grad + off[0]) = res[0];
                    if (upd) {
                        const float4 pv = adam_quad(off[0], res[0], pq[0], mq[0], vq[0]);
                        if (it == 0) W2_STAMP(13);
                        if (ad.shadow) {
                            const int nt = stab.n;
    #pragma unroll 1
                            for (int t = 0; t < nt; ++t) {
                                if (!((tmask >> t) & 1)) continue;
                                shadow_store4_desc(ad.shadow, stab.d[t], (stab.flags & (2 << t)) != 0, off[0], pv);
                            }
                        }
                    }
                } else if (valid[0] != 0) {      // (a record the host should not have fused)
                    __hip_atomic_store(ad.health, ERC_HEALTH_RAISED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (it < 2) W2_STAMP(11 + it);
    #pragma unroll
                for (int o = 0; o < 7; ++o)
                    res[o] = res[o + 1], pq[o] = pq[o + 1], mq[o] = mq[o + 1], vq[o] = vq[o + 1], off[o] = off[o + 1], valid[o] = valid[o + 1];
            }
            if (bdst) {      // bias strips: split 0 of the tile
                const float v = bsum;
                *(ERC_GLOBAL float*)bdst = v;
                if (upd) {
                    ac.upd(bp, v, bm, bv);
                    ad.data[boff] = bp, ad.m[boff] = bm, ad.v[boff] = bv;
                    if (ad.shadow) {
                        const int nt = stab.n;
    #pragma unroll 1
                        for (int t = 0; t < nt; ++t) shadow_store_desc(ad.shadow, stab.d[t], boff, bp);
                    }
                }
            }
            W2_STAMP(10);
            return;
        }

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
    };
    if (!WIDE) {
        finish(tn, 1.f);
    } else {
        const int tg = 4 * (local / d.splits);
        for (int t = 0; t < 4 && tg + t < d.tiles_n; ++t) {
            if (t) __syncthreads();        // the previous tile's readers of red / bred / s_flag are done
            finish(tg + t, w == t ? 1.f : 0.f);
        }
    }
}

struct W2Bases {  // first work item of every descriptor, passed by value (no dependent table reads to find one's descriptor)
    int v[W2_MAX_DESC];
};

template <bool ADAM, bool WIDE, int NT>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const W2Desc* __restrict__ table, const int n_desc, const W2Bases bases,
                                                         float* slabs, int* counters, uint64_t* stamps, int stamp_item, const W2Adam ad) {
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
    const int local = L - bases.v[di];      // (WIDE: a workgroup covers four column tiles -- item_base counts slabs, bases workgroups)
    if (local >= d.n_items) return;
    // (stamp_item is a work item of record 0)
    const bool st_item = stamps && L == stamp_item, st_tile = stamps && di == 0 && local / d.splits == stamp_item / d.splits;
    w2_body<ADAM, WIDE, NT>(d, local, red, bred, idx, &s_flag, slabs, counters, st_item ? stamps : nullptr, st_tile ? stamps : nullptr, ad);
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

static int g_w2_spin_limit = 2000000;
// test hook: bound of the fused-optimizer launch's wait for a tile's splits (<= 0 restores the default).  With a bound of 1
// the first wait that is not satisfied at once raises the health word: the timeout path end to end (tests/test_gpu_cogmen.py).
extern "C" int erc_wgrad_bf16_set_spin_limit(int limit) {
    g_w2_spin_limit = limit > 0 ? limit : 2000000;
    return ERC_OK;
}

// table: n_desc W2Desc records (device memory, <= 16); item_base: HOST array of the records' item_base fields; n_items = sum
// of tiles * splits; slabs: n_items * erc_wgrad_bf16_slab_floats() floats; counters: one zero-initialised int32 per output
// tile (left zero by the launch).
static int w2_launch(int mode, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs, int32_t* counters,
                     const W2Adam& ad, void* stream, int terms = 1) {
    ERC_REQUIRE(table && item_base && n_desc > 0 && n_desc <= W2_MAX_DESC && n_items > 0 && slabs && counters,
                "wgrad_bf16: bad arguments (at most %d records per launch)", W2_MAX_DESC);
    W2Bases bases{};
    for (int t = 0; t < n_desc; ++t) {
        ERC_REQUIRE(item_base[t] >= 0 && item_base[t] < n_items && (t == 0 ? item_base[0] == 0 : item_base[t] > item_base[t - 1]),
                    "wgrad_bf16: item_base[%d] = %d", t, item_base[t]);
        bases.v[t] = item_base[t];
    }
    ERC_REQUIRE(terms >= 1 && terms <= 3 && (terms == 1 || mode != 2), "wgrad_split: terms = %d (1 .. 3; the wide form is bf16 only)", terms);
#define W2_GO(ADAM_, NT_)                                                                                                              \
    hipLaunchKernelGGL((wgrad_bf16_kernel<ADAM_, false, NT_>), dim3(n_items), dim3(256), 0, (hipStream_t)stream, (const W2Desc*)table, \
                       n_desc, bases, slabs, counters, g_w2_stamps, g_w2_stamp_item, ad)
    if (terms == 2) {
        if (mode == 1) W2_GO(true, 2);
        else W2_GO(false, 2);
    } else if (terms == 3) {
        if (mode == 1) W2_GO(true, 3);
        else W2_GO(false, 3);
    } else if (mode == 1)
        hipLaunchKernelGGL((wgrad_bf16_kernel<true, false, 1>), dim3(n_items), dim3(256), 0, (hipStream_t)stream, (const W2Desc*)table, n_desc,
                           bases, slabs, counters, g_w2_stamps, g_w2_stamp_item, ad);
    else if (mode == 2)
        hipLaunchKernelGGL((wgrad_bf16_kernel<false, true, 1>), dim3(n_items), dim3(256), 0, (hipStream_t)stream, (const W2Desc*)table, n_desc,
                           bases, slabs, counters, g_w2_stamps, g_w2_stamp_item, ad);
    else
        hipLaunchKernelGGL((wgrad_bf16_kernel<false, false, 1>), dim3(n_items), dim3(256), 0, (hipStream_t)stream, (const W2Desc*)table, n_desc,
                           bases, slabs, counters, g_w2_stamps, g_w2_stamp_item, ad);
#undef W2_GO
    ERC_LAUNCH_CHECK("wgrad_bf16");
    return ERC_OK;
}

extern "C" int erc_wgrad_bf16(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                              int32_t* counters, void* stream) {
    W2Adam ad{};
    return w2_launch(0, table, n_desc, item_base, n_items, slabs, counters, ad, stream);
}

// The same products for LARGE K (N = 33 k nodes at B = 512): a workgroup's four wavefronts take four neighbouring column tiles
// over the same k-steps instead of splitting K (see w2_body).  wg_base: HOST array, first workgroup of every record
// (ceil(tiles_n / 4) * splits workgroups per record); the records' item_base / n_items fields = first slab (tiles_n * splits
// slabs per record) / number of workgroups; n_wgs = total workgroups; slabs and counters as erc_wgrad_bf16.
extern "C" int erc_wgrad_bf16_wide(const void* table, int n_desc, const int32_t* wg_base, int n_wgs, float* slabs, int32_t* counters,
                                   void* stream) {
    W2Adam ad{};
    return w2_launch(2, table, n_desc, wg_base, n_wgs, slabs, counters, ad, stream);
}

// erc_wgrad_bf16 with the optimizer fused in (W2Adam above): every record's C / bias_a / bias_b must point into g[0, n);
// the same offsets of p / m / v receive torch.optim.Adam's (decoupled != 0: AdamW's) update with the finished gradient
// times grad_scale, the bf16 shadows of the table follow.  Records of kind 1 name ranges of g that an earlier launch
// completed.  counters: one int32 per output tile + 512 (the private launch sequence numbers), zero-filled ONCE and owned
// by this entry point (the per-tile counters count up monotonically here).  n_items <= 256: every work item resident.
static int w2_adam_entry(int terms, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                         int32_t* counters, int n_tiles, float* p, float* g, float* m, float* v, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                         int64_t* state, void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host,
                         int32_t* health, void* stream, const ErcP2P* x = nullptr) {
    static_assert(sizeof(ShadowTab) == sizeof(ErcShadowTab), "shadow table layout");
    ERC_REQUIRE(p && g && m && v && state && health && n > 0 && n_tiles >= 0 && n_items <= 256,
                "wgrad_bf16_adam: bad arguments (at most 256 work items: all of them must be resident)");
    ERC_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "wgrad_bf16_adam: 16-byte alignment");
    W2Adam ad{};
    ad.decoupled = decoupled, ad.lr = lr, ad.b1 = beta1, ad.b2 = beta2, ad.eps = eps, ad.wd = weight_decay;
    ad.grad_scale = grad_scale, ad.data = p, ad.grad = g, ad.m = m, ad.v = v, ad.state = state, ad.skip = health, ad.health = health;
    ad.seq = counters + n_tiles;
    ad.spin_limit = g_w2_spin_limit;
    if (x) {
        ERC_REQUIRE(x->world >= 2 && x->world <= 8 && x->rank >= 0 && x->rank < x->world && x->epoch && x->n_pad >= n && x->n_pad % 4 == 0 &&
                        (int32_t*)x->health == health, "wgrad_adam_p2p: bad exchange descriptor (2 <= world <= 8; health = the exchange's health word)");
        ad.x_world = x->world, ad.x_rank = x->rank, ad.x_spin = x->spin_limit > 0 ? x->spin_limit : 4000000;
        ad.x_epoch = (int64_t*)x->epoch, ad.x_npad = x->n_pad;
        for (int r = 0; r < x->world; ++r) {
            ERC_REQUIRE(x->pub[r] && x->flags[r] && ((uintptr_t)x->pub[r] & 15) == 0, "wgrad_adam_p2p: peer %d not mapped", r);
            ad.x_pub[r] = (float*)x->pub[r], ad.x_flags[r] = (int32_t*)x->flags[r];
        }
    }
    if (tab_host) memcpy(&ad.tab, tab_host, sizeof(ad.tab));
    ERC_REQUIRE(ad.tab.n == 0 || shadow_base, "wgrad_bf16_adam: shadow table without a shadow buffer");
    ERC_REQUIRE(ad.tab.n >= 0 && ad.tab.n <= SHADOW_MAX, "wgrad_bf16_adam: %d shadow descriptors", ad.tab.n);
    // the quad fast path flags of erc_adam_step_tab (bit 0: every range on quads; bit 1 + t: range t runs along k)
    ad.tab.flags = ad.tab.n > 0;
    for (int t = 0; t < ad.tab.n; ++t) {
        const ShadowDesc& d = ad.tab.d[t];
        ERC_REQUIRE(d.src_off >= 0 && d.n_el > 0 && d.src_off + d.n_el <= n && d.n0 > 0 && d.n1 > 0 && d.ld > 0,
                    "wgrad_bf16_adam: shadow descriptor %d out of range", t);
        if (d.src_off % 4 || d.n0 % 4 || d.n_el % 4) ad.tab.flags &= ~1;
        const bool kq = d.sn0 == 0 && d.sk0 == 1 && d.sk1 % 4 == 0 && d.sk2 % 4 == 0 && d.dst_off % 4 == 0 && d.plane_stride % 4 == 0 &&
                        (d.mode == 1 || d.ld % 4 == 0) && ((uintptr_t)shadow_base & 7) == 0;
        if (kq) ad.tab.flags |= 2 << t;
    }
    (void)shadow_numel;     // (bounds-checked against the buffer by erc_shadow_refresh / erc_adam_step_tab when the table was built)
    ad.shadow = ad.tab.n > 0 ? (unsigned short*)shadow_base : nullptr;
    return w2_launch(1, table, n_desc, item_base, n_items, slabs, counters, ad, stream, terms);
}

extern "C" int erc_wgrad_bf16_adam(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                                   int32_t* counters, int n_tiles, float* p, float* g, float* m, float* v, int64_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                                   int64_t* state, void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host,
                                   int32_t* health, void* stream) {
    return w2_adam_entry(1, table, n_desc, item_base, n_items, slabs, counters, n_tiles, p, g, m, v, n, lr, beta1, beta2, eps,
                         weight_decay, decoupled, grad_scale, state, shadow_base, shadow_numel, tab_host, health, stream);
}

// SPLIT COMPUTE MODES (terms = 2 | 3; csrc/split_dev.h): the same launches for FP32 operands -- the records' A [K, lda] and
// B [K or gathered rows, ldb] are fp32 (lda, ldb multiples of 4, rows 16-byte aligned, M and N multiples of 4; columns
// [M, 8 ceil(M / 8)) of A are never read) -- expanded into `terms` bf16 terms in registers: fp32-class weight gradients at the
// bf16 matrix cores' rate.  Tables, slabs, counters and the fused optimizer exactly as erc_wgrad_bf16 / erc_wgrad_bf16_adam.
extern "C" int erc_wgrad_split(int terms, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                               int32_t* counters, void* stream) {
    ERC_REQUIRE(terms == 2 || terms == 3, "wgrad_split: terms = %d (2 or 3)", terms);
    W2Adam ad{};
    return w2_launch(0, table, n_desc, item_base, n_items, slabs, counters, ad, stream, terms);
}
extern "C" int erc_wgrad_split_adam(int terms, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                                    int32_t* counters, int n_tiles, float* p, float* g, float* m, float* v, int64_t n, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                                    int64_t* state, void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host,
                                    int32_t* health, void* stream) {
    ERC_REQUIRE(terms == 2 || terms == 3, "wgrad_split_adam: terms = %d (2 or 3)", terms);
    return w2_adam_entry(terms, table, n_desc, item_base, n_items, slabs, counters, n_tiles, p, g, m, v, n, lr, beta1, beta2, eps,
                         weight_decay, decoupled, grad_scale, state, shadow_base, shadow_numel, tab_host, health, stream);
}

// DATA PARALLEL with the exchange inside (ERC_DP_P2P=1; ErcP2P as erc_adam_step_p2p): erc_wgrad_bf16_adam (terms = 1) /
// erc_wgrad_split_adam (2 | 3) whose work items, having summed their quads over the tile's splits, publish them, rendezvous with
// the same item of every rank (bounded; health bits travel with the flags) and apply the update with the RANK-ORDERED sum times
// grad_scale (= 1 / world): the N > 1 step stays 5 launches, no RCCL call.  Every rank must launch the SAME table (same split
// counts: same work items).  The flag arrays hold world * 512 int32 (item index < 256); x->epoch: int64 [512], zero-filled once.
extern "C" int erc_wgrad_adam_p2p(int terms, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                                  int32_t* counters, int n_tiles, float* p, float* g, float* m, float* v, int64_t n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                                  int64_t* state, void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host,
                                  const ErcP2P* x, void* stream) {
    ERC_REQUIRE(x && terms >= 1 && terms <= 3, "wgrad_adam_p2p: bad arguments");
    return w2_adam_entry(terms, table, n_desc, item_base, n_items, slabs, counters, n_tiles, p, g, m, v, n, lr, beta1, beta2, eps,
                         weight_decay, decoupled, grad_scale, state, shadow_base, shadow_numel, tab_host, (int32_t*)x->health, stream, x);
}
