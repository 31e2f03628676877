// (Bi)LSTM recurrence, hidden size 100 per direction -- the sequence-context encoders of DialogueGCN
// (SeqContext, packed: track_mm/dgcn_models.py:10-33) and MMGCN (text branch, unpacked over the padded
// length: track_mm/mmgcn.py:69,113-114).  torch.nn.LSTM semantics, gate order i|f|g|o.
//
// One workgroup per (dialogue, direction).  The input-side gate pre-activations GX = x W_ih^T + b_ih are one
// hoisted GEMM over all rows (both directions: 800 columns); the recurrent matrix W_hh [400,100] of the
// direction lives in REGISTERS for the whole scan (one gate row per thread: 100 VGPRs), h_{t-1} is
// broadcast from LDS, so a step costs ~100 LDS broadcasts + 100 FMAs per thread and touches HBM only for the
// GX row (requested 4 steps ahead) and the saved state.  The backward scan keeps the transposed slices in registers the same way and
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
    unsigned long long* stamps;          // diagnostic (erc_lstm_set_stamps): shader-clock stamps of one step of workgroup (0,0)
};
unsigned long long* g_lstm_stamps = nullptr;
#define LSTM_STAMP(r_, slot)                                                                       \
    do {                                                                                           \
        if (r_ == 1 && s0 == SC && p.stamps && tid == 0 && b == 0 && d == 0) {                     \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            p.stamps[slot] = __builtin_readcyclecounter();                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    } while (0)

// rows of a dialogue: row(t) = base + t * step
struct RowMap {
    int64_t base, step;
    __device__ __forceinline__ int64_t operator()(int t) const { return base + (int64_t)t * step; }
};
__device__ __forceinline__ RowMap rows_of(const LstmP& p, int b) {
    return p.node_off ? RowMap{(int64_t)p.node_off[b], 1} : RowMap{(int64_t)b * p.sb, p.st};
}
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
// barrier that orders LDS traffic only: __syncthreads() would also wait for the LDS-DMA requests in flight (vmcnt)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// sigmoid / tanh on the hardware exponential and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each; absolute error < 3e-7):
// 4 and 6 instructions -- the library expf + IEEE division + tanhf were 90 of the 230 instructions of a step, and the
// step is bound by instruction issue (two wavefronts per SIMD, finding 26)
__device__ __forceinline__ float fast_sigm(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x));
}
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * fast_sigm(2.0f * x) - 1.0f; }

// value of lane Q of the caller's quad (lanes 4u .. 4u+3), in every lane of the quad: one DPP move, no LDS
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), Q * 0x55, 0xF, 0xF, true));
}
// Global memory is touched once per SC steps, not per step: the wavefront's memory counter (vmcnt) retires loads and
// stores in issue order, so a step that stores its results and then needs a prefetched operand waits for its own stores --
// one L2 round trip (~0.9 us) per step whatever the prefetch distance.  A chunk's operands are requested one chunk ahead
// and its results are kept in registers until the chunk ends; inside a chunk a step is LDS + ALU only.
constexpr int SC = 8;
// Vectors that every thread reads (h_{t-1}: 100 values, the gate gradients: 400) sit in LDS as chunks of 25 values padded
// to 28 (16-byte rows, chunks 28 banks apart: the distinct addresses of one wavefront read never share a bank).
constexpr int CHK = 25, CHP = 28;
constexpr int HP = 4 * CHP;        // hidden state
constexpr int DPP_ = 16 * CHP;     // gate gradients
__device__ __forceinline__ int chunk_pos(int k) { return (k / CHK) * CHP + k % CHK; }
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Thread layout of both scans: quad u = tid / 4 (< 100 live) owns hidden unit u, lane q = tid % 4 of the quad its gate q
// (i|f|g|o).  The four gates of a unit meet through DPP moves inside the quad, the cell state / the recurrent gradients
// live in registers (the same value in the 4 lanes), and the only LDS traffic is the vector every thread needs from all
// others -- h_{t-1} forward, the gate gradients backward -- double-buffered, so a step has ONE barrier.
__global__ __launch_bounds__(NTH) void lstm_fwd_kernel(LstmP p) {
    const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const int L = p.lengths ? (int)p.lengths[b] : p.T;
    const RowMap rmap = rows_of(p, b);
    __shared__ __attribute__((aligned(16))) float s_h[2][HP];
    const int u = tid >> 2, q = tid & 3;
    const bool live = u < H;
    const int uc = min(u, H - 1);
    const int grow = q * H + uc;       // the gate this thread activates: its column of GX / gates
    // the recurrent product: the quad splits K -- lane q multiplies the unit's FOUR gate rows by h[25q .. 25q+25), 100
    // weights in registers but 25 LDS values per step instead of 100 (every lane reading the whole vector was 200 KB of
    // LDS traffic a step), as packed fp32 multiply-adds over the gate pairs (i,f) and (g,o)
    f2 w01[CHK], w23[CHK];
    {
        const float* src = p.W_hh + ((int64_t)d * G4 + uc) * H + q * CHK;
#pragma unroll
        for (int i = 0; i < CHK; ++i) {
            w01[i] = f2{src[i], src[H * H + i]};
            w23[i] = f2{src[2 * H * H + i], src[3 * H * H + i]};
        }
    }
    const float bhh = p.b_hh[d * G4 + grow];
    if (tid < HP) s_h[0][tid] = 0.f, s_h[1][tid] = 0.f;
    uint64_t roff = 0, rseed = 0;
    const bool dropping = p.Hdrop && p.drop_p > 0.f;
    if (dropping) roff = p.rng[0], rseed = p.rng[1] ^ p.rng_stream;
    const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const float am = q == 2 ? 2.f : 1.f;       // tanh(a) = 2 sigm(2a) - 1: one exponential whatever the gate
    const bool odd = q & 1, hi = q & 2;
    float c = 0.f, hprev = 0.f;
    __syncthreads();
    const int gcol = d * G4 + grow;
    auto t_of = [&](int s) { return d == 0 ? s : L - 1 - s; };
    // row of scan step s = row_first + s * dstep: the chunk boundaries are instruction-count bound (a wavefront issues
    // one instruction per 4 cycles whatever its kind), so addresses are a uniform row pointer + a 32-bit lane offset, or
    // a per-lane pointer advanced by a per-lane step
    const int64_t dstep = d == 0 ? rmap.step : -rmap.step;
    const int64_t row_first = rmap(t_of(0));
    // the unit's four per-step outputs, one per lane of the quad: c | h_{t-1} | h | dropped h
    float* const obase = q == 0 ? p.Cst : q == 1 ? p.Hprev : q == 2 ? p.Hout : p.Hdrop;
    const int64_t opitch = q < 2 ? 2 * H : q == 2 ? p.ldh : p.ldhd;
    const bool ostore = live && (q < 3 || p.Hdrop);
    float* const uptr0 = obase + row_first * opitch + d * H + uc;
    const int64_t ustep = dstep * opitch;
    // A chunk's operands travel global -> LDS directly (LDS-DMA, lane l of a wavefront lands at base + 4 l): requested at
    // the previous chunk boundary, complete long before the next one, read back by the thread that asked -- no destination
    // registers, so nothing in the steps can wait for them.  (Prefetching into registers made the steps stall: carried
    // from boundary to boundary the compiler copies them behind the new requests, and while a request is pending any
    // packed multiply-add whose unused operand half happens to be the register next to it waits for the request.)
    __shared__ float s_gx[2][SC][NTH];
    float gx_cur[SC], o_gate[SC], o_unit[SC];
    // per-lane pointers advanced by uniform steps: a store or request is then ~6 instructions, not 30 - 50 of 64-bit row
    // arithmetic (steps past the end of the dialogue are skipped by a uniform branch; requests are clamped to the last row)
    const int64_t gx_step = dstep * p.ldgx, gate_step = dstep * (2 * G4);
    auto request_chunk = [&](int c0, int buf) {
        float* dst = &s_gx[buf][0][tid & ~63];
        const float* src = p.GX + (row_first + (int64_t)c0 * dstep) * p.ldgx + gcol;
        const float* last = p.GX + (row_first + (int64_t)(L - 1) * dstep) * p.ldgx + gcol;
#pragma unroll
        for (int r = 0; r < SC; ++r) {
            __builtin_amdgcn_global_load_lds(c0 + r < L ? src : last, (lds_void*)(dst + r * NTH), 4, 0, 0);
            src += gx_step;
        }
    };
    if (L > 0) request_chunk(0, 0);
    auto store_chunk = [&](int c0) {
        float* up = uptr0 + c0 * ustep;
        float* gp = p.gates + (row_first + (int64_t)c0 * dstep) * (2 * G4) + gcol;
        int64_t row = row_first + (int64_t)c0 * dstep;
#pragma unroll
        for (int r = 0; r < SC; ++r) {
            if (c0 + r < L) {       // uniform
                if (live) *gp = o_gate[r];
                float val = o_unit[r];
                if (dropping) {    // uniform
                    const float uu = erc_uniform(rseed, roff, (uint64_t)row * 2 * H + d * H + uc);
                    val = q < 3 ? val : uu >= p.drop_p ? val * keep_scale : 0.f;
                }
                if (ostore) *up = val;
            }
            up += ustep, gp += gate_step, row += dstep;
        }
    };
    unsigned long long t_c0 = 0, t_r0 = 0;
    if (p.stamps && tid == 0 && b == 0 && d == 0) t_c0 = __builtin_readcyclecounter(), t_r0 = __builtin_amdgcn_s_memrealtime();
    for (int s0 = 0; s0 < L; s0 += SC) {
        // chunk boundary: this chunk's operands out of LDS, the previous chunk's results to memory, the next chunk's
        // operands requested
        const int cb = (s0 / SC) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // requests and stores of the previous boundary: a chunk old
#pragma unroll
        for (int r = 0; r < SC; ++r) gx_cur[r] = s_gx[cb][r][tid];
        if (s0 > 0) store_chunk(s0 - SC);
        if (s0 + SC < L) request_chunk(s0 + SC, cb ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < SC; ++r) {
            const int s = s0 + r;
            if (s < L) {       // uniform
                const int cur = s & 1;
                LSTM_STAMP(r, 0);
                const float* hv = s_h[cur] + q * CHP;
                // two accumulator sets (even / odd k): four independent chains of packed multiply-adds
                f2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f}, r01 = {0.f, 0.f}, r23 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < CHK; ++i) {
                    const f2 hk = {hv[i], hv[i]};
                    if (i & 1) {
                        r01 = __builtin_elementwise_fma(w01[i], hk, r01);
                        r23 = __builtin_elementwise_fma(w23[i], hk, r23);
                    } else {
                        p01 = __builtin_elementwise_fma(w01[i], hk, p01);
                        p23 = __builtin_elementwise_fma(w23[i], hk, p23);
                    }
                }
                p01 += r01, p23 += r23;
                LSTM_STAMP(r, 1);
                // quad transpose-reduce: lane q ends with gate q's sum over the four lanes (3 DPP adds, no branches):
                // lanes exchange across xor 1 the gate of pair (0,1) / (2,3) they do not keep, then across xor 2
                const float k0 = odd ? p01.y : p01.x, g0 = odd ? p01.x : p01.y;
                const float k1 = odd ? p23.y : p23.x, g1 = odd ? p23.x : p23.y;
                const float e0 = k0 + dpp_mov<0xB1>(g0), e1 = k1 + dpp_mov<0xB1>(g1);
                const float a = gx_cur[r] + bhh + ((hi ? e1 : e0) + dpp_mov<0x4E>(hi ? e0 : e1));
                const float sg = fast_sigm(am * a);
                const float act = q == 2 ? 2.f * sg - 1.f : sg;
                LSTM_STAMP(r, 2);
                const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
                c = gf * c + gi * gg;
                const float h = go * fast_tanh(c);
                o_gate[r] = act;
                o_unit[r] = q == 0 ? c : q == 1 ? hprev : h;      // lane 3: h, dropped when it is stored
                if (live && q == 0) s_h[cur ^ 1][chunk_pos(u)] = h;
                hprev = h;
                LSTM_STAMP(r, 3);
                lds_barrier();
                LSTM_STAMP(r, 4);
            }
        }
    }
    if (p.stamps && tid == 0 && b == 0 && d == 0) {
        p.stamps[5] = __builtin_readcyclecounter() - t_c0;
        p.stamps[6] = __builtin_amdgcn_s_memrealtime() - t_r0;
        p.stamps[7] = L;
    }
    if (L > 0) store_chunk((L - 1) / SC * SC);
    // padded positions: zero output (pad_packed_sequence) -- only meaningful for padded row addressing
    if (!p.node_off)
        for (int t = L; t < p.T; ++t) {
            const int64_t row = rmap(t);
            if (tid < H) {
                p.Hout[row * p.ldh + d * H + tid] = 0.f;
                if (p.Hdrop) p.Hdrop[row * p.ldhd + d * H + tid] = 0.f;
            }
        }
}

__global__ __launch_bounds__(NTH) void lstm_bwd_kernel(LstmP p) {
    const int b = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
    const int L = p.lengths ? (int)p.lengths[b] : p.T;
    const RowMap rmap = rows_of(p, b);
    __shared__ __attribute__((aligned(16))) float s_dp[2][DPP_];     // gate gradients of the step, entry q*H + j chunked
    const int u = tid >> 2, q = tid & 3;
    const bool live = u < H;
    const int uc = min(u, H - 1);
    // the recurrent product (W_hh^T dpre)[u]: a DPP row of 16 lanes = 4 units; lane `part` of the row multiplies entries
    // [25 part, 25 part + 25) of the 400 gate gradients into each of the row's 4 units (100 weights in registers, 25 LDS
    // values per step), the row sums by DPP.  The elementwise part keeps the quad layout: unit u = tid / 4 is unit
    // (tid / 4) % 4 of its own row.
    const int part = tid & 15, u0 = min(tid >> 4, H / 4 - 1) * 4;
    f2 wt01[CHK], wt23[CHK];
#pragma unroll
    for (int i = 0; i < CHK; ++i) {
        const float* src = p.W_hh + ((int64_t)d * G4 + part * CHK + i) * H + u0;
        wt01[i] = f2{src[0], src[1]};
        wt23[i] = f2{src[2], src[3]};
    }
    const int myr = (tid >> 2) & 3;
    uint64_t roff = 0, rseed = 0;
    const bool dropped = p.drop_p > 0.f;
    if (dropped) roff = p.rng[0], rseed = p.rng[1] ^ p.rng_stream;
    const float keep_scale = dropped ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    auto t_of = [&](int s) { return d == 0 ? s : L - 1 - s; };
    // operands of a step: upstream gradient, the thread's own gate, the cell state (the previous cell state is the next
    // step's); clamped rows, a chunk ahead
    struct ChunkIn { float g[SC], gq[SC], c[SC + 1]; };      // steps s0, s0-1, ..; c has one more: c_prev of the last step
    const int64_t dstep = d == 0 ? rmap.step : -rmap.step;       // row of scan step s = row_first + s * dstep
    const int64_t row_first = rmap(t_of(0));
    const int lane_h = d * H + uc, lane_g = d * G4 + q * H + uc;
    auto load_chunk = [&](int s0, ChunkIn& X) {
#pragma unroll
        for (int r = 0; r <= SC; ++r) {
            const int64_t row = row_first + (int64_t)min(max(s0 - r, 0), max(L - 1, 0)) * dstep;     // uniform
            if (r < SC) {
                X.g[r] = (p.dHout + row * p.lddh)[lane_h];
                X.gq[r] = (p.gates + row * 2 * G4)[lane_g];
            }
            X.c[r] = (p.Cst + row * 2 * H)[lane_h];
        }
    };
    auto touch = [&](ChunkIn& X) {
#pragma unroll
        for (int r = 0; r <= SC; ++r) {
            if (r < SC) asm volatile("" : "+v"(X.g[r]), "+v"(X.gq[r]));
            asm volatile("" : "+v"(X.c[r]));
        }
    };
    float o_dp[SC];
    auto store_chunk = [&](int s0) {
#pragma unroll
        for (int r = 0; r < SC; ++r) {
            const int s = s0 - r;
            if (s >= 0 && live) (p.dGX + (row_first + (int64_t)s * dstep) * 2 * G4)[lane_g] = o_dp[r];
        }
    };
    float dh_rec = 0.f, dc_carry = 0.f;
    auto steps = [&](int s0, const ChunkIn& X) {
        // everything of a step that does not depend on the recurrent gradients, for the whole chunk: with
        //   P = o (1 - tanh(c)^2),  K = gate' x (g | c_prev | i | tanh(c)) for this lane's gate,
        // the dependent chain of a step is dh = g + dh_rec; dc = dc_carry + dh P; dpre = (o-lane ? dh : dc) K; dc_carry = dc f
        float fP[SC], fK[SC], fF[SC], fG[SC];
#pragma unroll
        for (int r = 0; r < SC; ++r) {
            const int s = max(s0 - r, 0);
            const float cprev = s > 0 ? X.c[r + 1] : 0.f;
            float g = X.g[r];
            if (dropped) {     // uniform
                const float uu = erc_uniform(rseed, roff, (uint64_t)rmap(t_of(s)) * 2 * H + d * H + uc);
                g = uu >= p.drop_p ? g * keep_scale : 0.f;
            }
            const float gq = X.gq[r];
            const float gi = quad_bcast<0>(gq), gf = quad_bcast<1>(gq), gg = quad_bcast<2>(gq), go = quad_bcast<3>(gq);
            const float tc = fast_tanh(X.c[r]);
            // d pre-activation of this lane's gate: i: dc g i(1-i) | f: dc c' f(1-f) | g: dc i (1-g^2) | o: dh tanh(c) o(1-o)
            const float der = q == 2 ? 1.f - gq * gq : gq * (1.f - gq);
            fP[r] = go * (1.f - tc * tc);
            fK[r] = der * (q == 0 ? gg : q == 1 ? cprev : q == 2 ? gi : tc);
            fF[r] = gf;
            fG[r] = g;
        }
#pragma unroll
        for (int r = 0; r < SC; ++r) {
            const int s = s0 - r;
            if (s >= 0) {      // uniform
                const int buf = s & 1;
                const float dh = fG[r] + dh_rec;
                const float dc = dc_carry + dh * fP[r];
                const float dp = (q == 3 ? dh : dc) * fK[r];
                dc_carry = dc * fF[r];
                o_dp[r] = dp;
                if (live) s_dp[buf][chunk_pos(q * H + u)] = dp;
                __syncthreads();
                const float* dv = s_dp[buf] + part * CHP;
                f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < CHK; ++i) {
                    const f2 dk = {dv[i], dv[i]};
                    a01 = __builtin_elementwise_fma(wt01[i], dk, a01);
                    a23 = __builtin_elementwise_fma(wt23[i], dk, a23);
                }
                float acc[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {   // sum over the 16 lanes of the row: quad xor 1 / xor 2, half mirror, mirror
                    acc[r4] += dpp_mov<0xB1>(acc[r4]);
                    acc[r4] += dpp_mov<0x4E>(acc[r4]);
                    acc[r4] += dpp_mov<0x141>(acc[r4]);
                    acc[r4] += dpp_mov<0x140>(acc[r4]);
                }
                dh_rec = myr == 0 ? acc[0] : myr == 1 ? acc[1] : myr == 2 ? acc[2] : acc[3];
            }
        }
    };
    ChunkIn inA, inB;
    if (L > 0) load_chunk(L - 1, inA);
    for (int s0 = L - 1; s0 >= 0; s0 -= 2 * SC) {
        __builtin_amdgcn_sched_barrier(0);
        touch(inA);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 < L - 1) store_chunk(s0 + SC);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 - SC >= 0) load_chunk(s0 - SC, inB);      // only requests that will be consumed (DESIGN.md finding 30)
        __builtin_amdgcn_sched_barrier(0);
        steps(s0, inA);
        if (s0 - SC >= 0) {
            __builtin_amdgcn_sched_barrier(0);
            touch(inB);
            __builtin_amdgcn_sched_barrier(0);
            store_chunk(s0);
            __builtin_amdgcn_sched_barrier(0);
            if (s0 - 2 * SC >= 0) load_chunk(s0 - 2 * SC, inA);
            __builtin_amdgcn_sched_barrier(0);
            steps(s0 - SC, inB);
        }
    }
    if (L > 0) store_chunk(L - 1 - (L - 1) / SC * SC);
    if (!p.node_off)
        for (int t = L; t < p.T; ++t) {
            const int64_t row = rmap(t);
            if (tid < G4) p.dGX[row * 2 * G4 + d * G4 + tid] = 0.f;
        }
}

}  // namespace

// diagnostic: shader-clock stamps inside one step of the next forward scans (tools/lstm_bench.py); nullptr = off
extern "C" int erc_lstm_set_stamps(unsigned long long* stamps) {
    g_lstm_stamps = stamps;
    return ERC_OK;
}

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
    p.stamps = g_lstm_stamps;
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
