// Batched weight gradients: every dW = A^T B product of a training step (A [K, M] and B [K, N] both K-major, K = #nodes)
// as ONE launch over a device-resident descriptor table.
//
// Replaces the autograd weight-gradient GEMMs behind loss.backward() (reference: track_mm/cogmen.py:187-189 and the
// other train_step bodies).
//
// Wave tile 64 x 64.  A K-major operand has its m (resp. n) index contiguous in memory, so one 16-byte load gives a
// lane FOUR neighbouring rows of the output for one k:  lane (r = lane & 15, g = lane >> 4) loads
// A[k0 + g][m0 + 4r .. 4r+3] and B[k0 + g][n0 + 4r .. 4r+3].  v_mfma_f32_16x16x4_f32 number (i, j) takes a[i] and
// b[j]: its 16 x 16 result holds the output rows {m0 + 4r' + i} and columns {n0 + 4r + j} -- a fixed permutation of
// the tile that is undone when the tile is stored.  2 loads feed 16 MFMAs (the 16 x 32 streaming kernel needs 12 for
// 8), and the bf16 feature block (input projection gradient) is read 8 bytes per lane, fully coalesced, with its row
// gather staged once per workgroup in LDS.
//
// K is split over the 4 wavefronts of a workgroup (reduced through LDS) and over `splits` workgroups per tile:
// partial tiles go to a slab, the workgroup that arrives last (agent-scope counter) sums the slabs in split order
// and writes the gradient -- no float atomics, bit-reproducible, no separate reduce launch.
#include "erc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Pointers read from the descriptor table are generic to the compiler: without the explicit global address space it
// emits flat_load, which also counts on lgkmcnt and so serialises with the LDS reads of the gather stage.
#define ERC_GLOBAL __attribute__((address_space(1)))
typedef const ERC_GLOBAL float* gfloat_cp;
typedef const ERC_GLOBAL unsigned short* gushort_cp;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef const ERC_GLOBAL f32x4* gfloat4_cp;
typedef const ERC_GLOBAL u16x4* gushort4_cp;

constexpr int WG_U = 2;            // k-steps (of 4 k each) per load batch; 2 keeps 3 workgroups per CU (146 VGPRs)
constexpr int WG_IDX_CAP = 2048;   // k per split (row-gather stage in LDS)
constexpr int WG_SLAB = 4096 + 64; // floats per partial tile: 64 x 64 + one bias strip
constexpr int WG_MAX_DESC = 32;

struct WgDesc {  // 112 bytes; mirrored by engine.GemmPlanner.flush_wgrads ("<QQQQQ14ifii4x")
    const float* A;
    const void* B;
    float* C;
    float* bias_out;          // ones == 1: [M] column sums of A;  ones == 2: [N] column sums of B
    const int32_t* b_gather;  // row of B for every k, or null
    int lda, ldb, ldc, M, N, K;
    int ones, b_bf16, splits, tiles_n, item_base, n_items, tile_base, vec;  // vec: bit0 A, bit1 B, bit2 C 16-byte ok
    float scale;              // the product is stored as scale * A^T B (GCNII: dW_l = theta_l dV_l); bias strips are not scaled
    int a_bf16;               // A is bf16 (the relation-mean tile the fused COGMEN forward stores); lda in elements
    int mma_bf16;             // bf16 compute mode: both operands rounded to bf16, v_mfma_f32_16x16x32_bf16 (fp32 accumulate; bias strips stay fp32 sums)
};

// Cross-workgroup hand-off of the partial tiles WITHOUT fences (an agent-scope release/acquire fence pair costs
// ~3.5 us per workgroup on MI355X: L2 write-back + invalidate): the slab is written with write-through (sc1) stores,
// one full 256-byte run per wave instruction, drained with s_waitcnt vmcnt(0) before the workgroup's arrival
// ticket; the last arriver reads it with sc1 loads.
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// VEC (16-byte operand loads legal for BOTH operands) is compile-time: a runtime flag around a load makes hipcc
// branch and drain vmcnt per load.
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// MB: bf16 matrix cores.  v_mfma_f32_16x16x32_bf16 sums over 32 (g, slot) pairs, 8 slots per lane; the load pattern
// stays the one above (one 16-byte load = four neighbouring output rows for ONE k), so a lane fills its 8 slots from 8
// consecutive k-steps of its wavefront -- slot j of lane (r, g) holds k-step j's k = 4 ks + g for both operands, which is
// all the instruction needs (any assignment of k to slots works as long as A and B agree).  16 MFMAs of 16 cycles per 8
// k-steps instead of 128 of 32 cycles; the accumulator layout, and with it the whole epilogue, is unchanged.
// MB == 2: fp32-class products on the bf16 matrix cores.  Every fp32 operand value x is split into three bf16 terms
// x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24 significant bits) and the six cross products of
// weight >= 2^-16 -- l l', m l' and l m' are below fp32 resolution -- are accumulated in fp32: 6 x 16 cycles per 32 k
// against 8 x 32 cycles of v_mfma_f32_16x16x4_f32, at fp32-class error (MMGCN's weight gradients ran at 76 % of the fp32
// matrix peak: the fp32 instruction itself was the ceiling).  Same loads, same accumulator layout, same epilogue.
template <bool ABF, bool BF16, bool VEC, int MB>
__device__ __forceinline__ void wgrad_body(const WgDesc& d, const int local, float* red, float* bred, int* idx,
                                           int* s_flag, float* slabs, int* counters) {
    const gfloat_cp A = (gfloat_cp)d.A;
    const ERC_GLOBAL int32_t* const gather = (const ERC_GLOBAL int32_t*)d.b_gather;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int split = local % d.splits, tile = local / d.splits;
    const int tm = tile / d.tiles_n, tn = tile % d.tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;
    const int nks = (d.K + 3) >> 2;
    const int per = (nks + d.splits - 1) / d.splits;
    const int ks_begin = split * per, ks_end = min(nks, ks_begin + per);
    const int k_begin = ks_begin * 4;
    const int nk = max(0, min(d.K, ks_end * 4) - k_begin);
    for (int t = tid; t < nk; t += 256) idx[t] = gather ? gather[k_begin + t] : k_begin + t;
    __syncthreads();

    constexpr bool avec = VEC, bvec = VEC;
    const int ma = m0 + 4 * r, nb = n0 + 4 * r;
    int mc[4], nc[4];
    bool m_ok[4], n_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m_ok[i] = ma + i < d.M, n_ok[i] = nb + i < d.N;
        mc[i] = min(ma + i, d.M - 1), nc[i] = min(nb + i, d.N - 1);
    }
    const int mav = m_ok[3] ? ma : 0, nbv = n_ok[3] ? nb : 0;  // vector paths: M, N multiples of 4 (host contract)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsa[4] = {0.f, 0.f, 0.f, 0.f}, bsb[4] = {0.f, 0.f, 0.f, 0.f};

    // Every global load is unconditional and nothing SELECTS on its result (hipcc sinks a load under the select's
    // condition: exec-masked branch + s_waitcnt per load).  Out-of-range k: address clamped, value multiplied by 0
    // (the clamped row is real data, so a NaN there is a NaN of the true result too).  Out-of-range m / n: address
    // clamped, the garbage only reaches output rows / columns that are never stored.
    // (the multiply is done in mma(), after the NEXT batch's loads have been issued: placed here it would make the
    // compiler wait for this batch right away)
    auto load = [&](const int ks, float (&a)[4], float (&b)[4], float& kf) {
        const int k = 4 * ks + g;
        kf = (ks < ks_end && k < d.K) ? 1.f : 0.f;
        const int kl = max(0, min(k - k_begin, nk - 1));
        const int64_t arow = (int64_t)(k_begin + kl) * d.lda;
        const int64_t brow = (int64_t)idx[kl] * d.ldb;
        if (ABF) {
            const gushort_cp Ah = (gushort_cp)d.A;
            unsigned short h[4];
            if (avec) {
                const u16x4 v = *(gushort4_cp)(Ah + arow + mav);
                h[0] = v.x, h[1] = v.y, h[2] = v.z, h[3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = Ah[arow + mc[i]];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(float, (unsigned)h[i] << 16);
        } else if (avec) {
            const f32x4 v = *(gfloat4_cp)(A + arow + mav);
            a[0] = v.x, a[1] = v.y, a[2] = v.z, a[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = A[arow + mc[i]];
        }
        if (BF16) {
            const gushort_cp Bh = (gushort_cp)d.B;
            unsigned short h[4];
            if (bvec) {
                const u16x4 v = *(gushort4_cp)(Bh + brow + nbv);
                h[0] = v.x, h[1] = v.y, h[2] = v.z, h[3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = Bh[brow + nc[j]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = __builtin_bit_cast(float, (unsigned)h[j] << 16);
        } else {
            const gfloat_cp Bf = (gfloat_cp)d.B;
            if (bvec) {
                const f32x4 v = *(gfloat4_cp)(Bf + brow + nbv);
                b[0] = v.x, b[1] = v.y, b[2] = v.z, b[3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bf[brow + nc[j]];
            }
        }
    };
    auto mma = [&](const float (&a0)[4], const float (&b0)[4], const float kf) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = a0[i] * kf, b[i] = b0[i] * kf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bsa[i] += a[i], bsb[i] += b[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    // wavefront w takes k-steps ks_begin + w + 4 s; batches of WG_U steps, the next batch's loads issued before
    // the current batch's MFMAs
    const int ns = (max(0, ks_end - ks_begin) + 3) >> 2;
    if (MB == 2) {
        auto split3 = [](float x, float& h, float& m, float& l) {
            h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);      // truncation: the remainder is exact
            const float r = x - h;
            m = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r) & 0xffff0000u);
            l = r - m;                                                                           // rounded to bf16 when packed
        };
        auto hi16 = [](float lo, float hi) -> unsigned {      // two values that already are bf16: take the high halves
            return (__builtin_bit_cast(unsigned, lo) >> 16) | (__builtin_bit_cast(unsigned, hi) & 0xffff0000u);
        };
        for (int s0 = 0; s0 < ns; s0 += 8) {
            float xa[8][4], xb[8][4], xk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) load(ks_begin + w + 4 * (s0 + u), xa[u], xb[u], xk[u]);
            __builtin_amdgcn_sched_barrier(0);
            u32x4 ah[4], am[4], al[4], bh[4], bm[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    float h0, m0, l0, h1, m1, l1;
                    const float x0 = xa[2 * dd][i] * xk[2 * dd], x1 = xa[2 * dd + 1][i] * xk[2 * dd + 1];
                    bsa[i] += x0 + x1;
                    split3(x0, h0, m0, l0), split3(x1, h1, m1, l1);
                    ah[i][dd] = hi16(h0, h1), am[i][dd] = hi16(m0, m1), al[i][dd] = pack_bf16x2(l0, l1);
                    const float y0 = xb[2 * dd][i] * xk[2 * dd], y1 = xb[2 * dd + 1][i] * xk[2 * dd + 1];
                    bsb[i] += y0 + y1;
                    split3(y0, h0, m0, l0), split3(y1, h1, m1, l1);
                    bh[i][dd] = hi16(h0, h1), bm[i][dd] = hi16(m0, m1), bl[i][dd] = pack_bf16x2(l0, l1);
                }
            }
#define ERC_MF(A_, B_) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_[i]), __builtin_bit_cast(bf16x8, B_[j]), acc[i][j], 0, 0, 0)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {      // small terms first
                    ERC_MF(al, bh);
                    ERC_MF(ah, bl);
                    ERC_MF(am, bm);
                    ERC_MF(am, bh);
                    ERC_MF(ah, bm);
                    ERC_MF(ah, bh);
                }
#undef ERC_MF
        }
    } else if (MB) {
        // groups of 8 steps = one MFMA per (i, j); loaded four steps at a time (the second half's loads are in flight
        // while the first half is rounded and packed)
        for (int s0 = 0; s0 < ns; s0 += 8) {
            float xa[4][4], xb[4][4], xk[4], ya[4][4], yb[4][4], yk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) load(ks_begin + w + 4 * (s0 + u), xa[u], xb[u], xk[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) load(ks_begin + w + 4 * (s0 + 4 + u), ya[u], yb[u], yk[u]);
            __builtin_amdgcn_sched_barrier(0);
            u32x4 fa[4], fb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xa[u][i] *= xk[u], xb[u][i] *= xk[u];
                    bsa[i] += xa[u][i], bsb[i] += xb[u][i];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i][0] = pack_bf16x2(xa[0][i], xa[1][i]), fa[i][1] = pack_bf16x2(xa[2][i], xa[3][i]);
                fb[i][0] = pack_bf16x2(xb[0][i], xb[1][i]), fb[i][1] = pack_bf16x2(xb[2][i], xb[3][i]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ya[u][i] *= yk[u], yb[u][i] *= yk[u];
                    bsa[i] += ya[u][i], bsb[i] += yb[u][i];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i][2] = pack_bf16x2(ya[0][i], ya[1][i]), fa[i][3] = pack_bf16x2(ya[2][i], ya[3][i]);
                fb[i][2] = pack_bf16x2(yb[0][i], yb[1][i]), fb[i][3] = pack_bf16x2(yb[2][i], yb[3][i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                                        acc[i][j], 0, 0, 0);
        }
    } else if (ns > 0) {
        // software pipeline without a conditional around any load (hipcc drains vmcnt at the end of a conditional
        // block that holds loads): batch b+1 is loaded unconditionally, batch b multiplied, registers rotated
        float xa[WG_U][4], xb[WG_U][4], xk[WG_U];
#pragma unroll
        for (int u = 0; u < WG_U; ++u) load(ks_begin + w + 4 * u, xa[u], xb[u], xk[u]);
        for (int s0 = WG_U; s0 < ns; s0 += WG_U) {
            float ya[WG_U][4], yb[WG_U][4], yk[WG_U];
#pragma unroll
            for (int u = 0; u < WG_U; ++u) load(ks_begin + w + 4 * (s0 + u), ya[u], yb[u], yk[u]);
            __builtin_amdgcn_sched_barrier(0);  // keep the loads in front of the MFMAs
#pragma unroll
            for (int u = 0; u < WG_U; ++u) mma(xa[u], xb[u], xk[u]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                xk[u] = yk[u];
#pragma unroll
                for (int i = 0; i < 4; ++i) xa[u][i] = ya[u][i], xb[u][i] = yb[u][i];
            }
        }
#pragma unroll
        for (int u = 0; u < WG_U; ++u) mma(xa[u], xb[u], xk[u]);
    }

    // bias strips: sum over the 4 k-slots g, then over wavefronts (below)
    const bool want_a = d.ones == 1 && tn == 0, want_b = d.ones == 2 && tm == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = want_b ? bsb[i] : bsa[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) bred[w * 64 + 4 * r + i] = v;
    }

    float* const slab = slabs + (int64_t)(d.item_base + local) * WG_SLAB;
    const bool direct = d.splits == 1;
    const bool cvec = d.vec & 4;
    auto store_c = [&](const int f4, const int h, const float4 v0) {
        const float4 v = make_float4(v0.x * d.scale, v0.y * d.scale, v0.z * d.scale, v0.w * d.scale);
        const int ilq = f4 >> 6, ln = f4 & 63;
        const int i = 2 * h + (ilq >> 2), q = ilq & 3;
        const int m = m0 + 4 * (4 * (ln >> 4) + q) + i, n = n0 + 4 * (ln & 15);
        if (m >= d.M || n >= d.N) return;
        ERC_GLOBAL float* dst = (ERC_GLOBAL float*)d.C + (int64_t)m * d.ldc + n;
        if (cvec) {
            *(ERC_GLOBAL f32x4*)dst = (f32x4){v.x, v.y, v.z, v.w};
        } else {
            dst[0] = v.x;
            if (n + 1 < d.N) dst[1] = v.y;
            if (n + 2 < d.N) dst[2] = v.z;
            if (n + 3 < d.N) dst[3] = v.w;
        }
    };
    // in-workgroup reduction in two passes (output-row tiles i = 2h, 2h+1): 32 KB of LDS
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();  // pass 0: idx[] reads finished (red aliases nothing, but bred must be complete); pass 1: red reuse
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 2 * h + il;
                const float4 v = make_float4(acc[i][0][q], acc[i][1][q], acc[i][2][q], acc[i][3][q]);
                *reinterpret_cast<float4*>(red + w * 2048 + ((il * 4 + q) * 64 + lane) * 4) = v;
            }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f4 = tid + 256 * u;
            float4 s = *reinterpret_cast<const float4*>(red + f4 * 4);
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const float4 t = *reinterpret_cast<const float4*>(red + ww * 2048 + f4 * 4);
                s.x += t.x, s.y += t.y, s.z += t.z, s.w += t.w;
            }
            if (direct) {
                store_c(f4, h, s);
            } else {  // slab layout [h][u][component][thread]
                float* q = slab + ((h * 2 + u) * 4) * 256 + tid;
                st_sc1(q, s.x), st_sc1(q + 256, s.y), st_sc1(q + 512, s.z), st_sc1(q + 768, s.w);
            }
        }
    }
    if ((want_a || want_b) && tid < 64) {
        const float v = bred[tid] + bred[64 + tid] + bred[128 + tid] + bred[192 + tid];
        if (direct) {
            const int c = (want_a ? m0 : n0) + tid;
            if (c < (want_a ? d.M : d.N)) ((ERC_GLOBAL float*)d.bias_out)[c] = v;
        } else {
            st_sc1(slab + 4096 + tid, v);
        }
    }
    if (direct) return;

    // publish the partial tile; the workgroup that arrives last reduces the slabs in split order
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* const counter = counters + d.tile_base + tile;
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == d.splits - 1;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;
    // 8 splits x 4 components in flight per thread (one dependent round trip per split would cost ~1 us each)
    const float* const tile_slabs = slabs + (int64_t)(d.item_base + tile * d.splits) * WG_SLAB;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int sp0 = 0; sp0 < d.splits; sp0 += 8) {
                float v[8][4], mk[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    mk[j] = sp0 + j < d.splits ? 1.f : 0.f;
                    const float* p = tile_slabs + (int64_t)min(sp0 + j, d.splits - 1) * WG_SLAB + ((h * 2 + u) * 4) * 256 + tid;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[j][c] = ld_sc1(p + 256 * c);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s.x += v[j][0] * mk[j], s.y += v[j][1] * mk[j], s.z += v[j][2] * mk[j], s.w += v[j][3] * mk[j];
            }
            store_c(tid + 256 * u, h, s);
        }
    if ((want_a || want_b) && tid < 64) {
        float v = 0.f;
        for (int sp0 = 0; sp0 < d.splits; sp0 += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = ld_sc1(tile_slabs + (int64_t)min(sp0 + j, d.splits - 1) * WG_SLAB + 4096 + tid);
#pragma unroll
            for (int j = 0; j < 8; ++j) v += t[j] * (sp0 + j < d.splits ? 1.f : 0.f);
        }
        const int c = (want_a ? m0 : n0) + tid;
        if (c < (want_a ? d.M : d.N)) ((ERC_GLOBAL float*)d.bias_out)[c] = v;
    }
}

// Which descriptor a work item belongs to, passed by value (no dependent table reads to find one's descriptor).  A RUN is a
// stretch of consecutive descriptors with the same number of work items (MMGCN: the 128 weight gradients of the GCNII chain are
// one run), so a launch takes WG_MAX_DESC runs, not WG_MAX_DESC descriptors -- MMGCN's 133 records were five launches, each with
// its own partly filled last round of work items.
struct WgRuns {
    int first_item[WG_MAX_DESC];   // first work item of the run
    int first_desc[WG_MAX_DESC];   // its first descriptor (index into the whole table)
    int per[WG_MAX_DESC];          // work items per descriptor
};

template <bool X3>
__device__ __forceinline__ void wgrad_dispatch(const WgDesc* __restrict__ table, const int n_runs, const WgRuns& runs,
                                               const int item_offset, float* slabs, int* counters, float* red, float* bred, int* idx,
                                               int* s_flag) {
    const int L = blockIdx.x + item_offset;
    int ri = 0;
#pragma unroll
    for (int t = 1; t < WG_MAX_DESC; ++t)
        if (t < n_runs && L >= runs.first_item[t]) ri = t;
    const int di = runs.first_desc[ri] + (L - runs.first_item[ri]) / runs.per[ri];
    const WgDesc d = table[di];
    const int local = L - d.item_base;
    if (local >= d.n_items) return;
    const bool vec = (d.vec & 3) == 3;
    if (X3 && d.mma_bf16 == 2 && !d.a_bf16 && !d.b_bf16 && vec) {   // fp32 operands, three-term bf16 split
        wgrad_body<false, false, true, 2>(d, local, red, bred, idx, s_flag, slabs, counters);
    } else if (d.mma_bf16 == 1) {   // bf16 matrix cores (COGMEN bf16 compute mode); vector access is a host contract there
        if (d.a_bf16) wgrad_body<true, false, true, 1>(d, local, red, bred, idx, s_flag, slabs, counters);
        else if (d.b_bf16) wgrad_body<false, true, true, 1>(d, local, red, bred, idx, s_flag, slabs, counters);
        else wgrad_body<false, false, true, 1>(d, local, red, bred, idx, s_flag, slabs, counters);
    } else if (d.a_bf16) {   // bf16 A with an fp32 B (COGMEN bf16 mode: d[W_r ; W_root] = M^T dH1)
        if (vec) wgrad_body<true, false, true, 0>(d, local, red, bred, idx, s_flag, slabs, counters);
        else wgrad_body<true, false, false, 0>(d, local, red, bred, idx, s_flag, slabs, counters);
    } else if (d.b_bf16) {
        if (vec) wgrad_body<false, true, true, 0>(d, local, red, bred, idx, s_flag, slabs, counters);
        else wgrad_body<false, true, false, 0>(d, local, red, bred, idx, s_flag, slabs, counters);
    } else {
        if (vec) wgrad_body<false, false, true, 0>(d, local, red, bred, idx, s_flag, slabs, counters);
        else wgrad_body<false, false, false, 0>(d, local, red, bred, idx, s_flag, slabs, counters);
    }
}

__global__ __launch_bounds__(256, 3) void wgrad_table_kernel(const WgDesc* __restrict__ table, const int n_desc,
                                                          const WgRuns bases, const int item_offset, float* slabs,
                                                          int* counters) {
    __shared__ float red[4 * 2048];
    __shared__ float bred[4 * 64];
    __shared__ int idx[WG_IDX_CAP];
    __shared__ int s_flag;
    wgrad_dispatch<false>(table, n_desc, bases, item_offset, slabs, counters, red, bred, idx, &s_flag);
}

// The same launch with the three-term bf16 split available (records with mma_bf16 == 2): its operand fragments need ~220
// registers, i.e. two workgroups per CU instead of three.
__global__ __launch_bounds__(256, 2) void wgrad_table_x3_kernel(const WgDesc* __restrict__ table, const int n_desc,
                                                             const WgRuns bases, const int item_offset, float* slabs,
                                                             int* counters) {
    __shared__ float red[4 * 2048];
    __shared__ float bred[4 * 64];
    __shared__ int idx[WG_IDX_CAP];
    __shared__ int s_flag;
    wgrad_dispatch<true>(table, n_desc, bases, item_offset, slabs, counters, red, bred, idx, &s_flag);
}

}  // namespace

extern "C" int64_t erc_wgrad_slab_floats(void) { return WG_SLAB; }
extern "C" int erc_wgrad_max_k_per_split(void) { return WG_IDX_CAP; }

// table: n_desc WgDesc records (device memory); item_base: HOST array of the records' item_base fields; n_items = sum of tiles * splits; slabs: n_items * erc_wgrad_slab_floats()
// floats; counters: one zero-initialised int32 per output tile (left zero by the launch).
static int wgrad_table_launch(bool x3, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                              int32_t* counters, void* stream) {
    ERC_REQUIRE(table && item_base && n_desc > 0 && n_items > 0 && slabs && counters, "wgrad_table: bad arguments");
    for (int t = 0; t < n_desc; ++t)
        ERC_REQUIRE(item_base[t] >= 0 && item_base[t] < n_items && (t == 0 ? item_base[0] == 0 : item_base[t] > item_base[t - 1]),
                    "wgrad_table: item_base[%d] = %d", t, item_base[t]);
    // the kernel finds a work item's descriptor from by-value runs: at most WG_MAX_DESC runs per launch
    int t = 0;
    while (t < n_desc) {
        WgRuns runs{};
        int nr = 0;
        const int first = item_base[t];
        while (t < n_desc && nr < WG_MAX_DESC) {
            const int per = (t + 1 < n_desc ? item_base[t + 1] : n_items) - item_base[t];
            runs.first_item[nr] = item_base[t], runs.first_desc[nr] = t, runs.per[nr] = per;
            ++t;
            while (t < n_desc && (t + 1 < n_desc ? item_base[t + 1] : n_items) - item_base[t] == per) ++t;
            ++nr;
        }
        const int end = t < n_desc ? item_base[t] : n_items;
        if (x3)
            hipLaunchKernelGGL(wgrad_table_x3_kernel, dim3(end - first), dim3(256), 0, (hipStream_t)stream, (const WgDesc*)table, nr, runs,
                               first, slabs, counters);
        else
            hipLaunchKernelGGL(wgrad_table_kernel, dim3(end - first), dim3(256), 0, (hipStream_t)stream, (const WgDesc*)table, nr, runs,
                               first, slabs, counters);
        ERC_LAUNCH_CHECK("wgrad_table");
    }
    return ERC_OK;
}

extern "C" int erc_wgrad_table(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                               int32_t* counters, void* stream) {
    return wgrad_table_launch(false, table, n_desc, item_base, n_items, slabs, counters, stream);
}
// records with mma_bf16 == 2 (fp32 operands, both 16-byte accessible) run as three-term bf16 splits: fp32-class products at
// ~2x the rate of the fp32 matrix-core instruction; every other record as in erc_wgrad_table
extern "C" int erc_wgrad_table_x3(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                                  int32_t* counters, void* stream) {
    return wgrad_table_launch(true, table, n_desc, item_base, n_items, slabs, counters, stream);
}
