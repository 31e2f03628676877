// S4: fused optimizer over the flat live-parameter buffer, and the library's
// error plumbing.
#include <stdarg.h>
#include <string.h>

#include "erc_common.h"
#include "optim_dev.h"

static thread_local char g_err[512] = "";

extern "C" void erc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* erc_last_error(void) { return g_err; }
extern "C" int erc_abi_version(void) { return ERC_ABI_VERSION; }

namespace {

__global__ __launch_bounds__(256) void shadow_refresh_kernel(const float* __restrict__ p, unsigned short* __restrict__ shadow,
                                                             const ShadowTab tab) {
    const ShadowDesc& d = tab.d[blockIdx.y];
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < d.n_el; idx += (int64_t)gridDim.x * 256) {
        const int32_t x = (int32_t)idx;
        const int32_t q = x / d.n0, d0 = x - q * d.n0, d2 = q / d.n1, d1 = q - d2 * d.n1;
        const int64_t dst = shadow_dst(d, d0 * d.sn0 + d1 * d.sn1 + d2 * d.sn2, d0 * d.sk0 + d1 * d.sk1 + d2 * d.sk2);
        float r = p[d.src_off + idx];
        for (int t = 0; t < d.terms; ++t) shadow[dst + t * d.plane_stride] = bf_term_next(r);
    }
}

// ---- one-shot gradient exchange fused into the optimizer launch (ERC_DP_P2P=1; SURVEY.md 8e: the latency-class
//      all-reduce for COGMEN's 1.1 MB of gradients).  Every rank maps every peer's PUBLISH buffer and FLAG array (hipIpc);
//      workgroup b of a rank owns quads [256 b, 256 b + 256) of the flat gradient:
//        1. it copies its chunk of the local gradient into the publish buffer of this step's parity (write-through,
//           system scope), drains, and stores (step << 1 | local health bit) into flags[rank][b] of EVERY rank (one 4-byte
//           store per peer over xGMI);
//        2. it polls its LOCAL flags[r][b], r = 0 .. world-1 (bounded; a timeout raises the health word), i.e. it waits for
//           the same chunk of every peer -- no grid-wide barrier, chunks pipeline;
//        3. it sums the chunk of all ranks IN RANK ORDER (its own included, read back from the publish buffer): the sum is
//           bit-identical on every rank; the update follows with grad_scale = 1 / world.
//      The parity double-buffers the publish area: a peer can only be one step ahead (it needs this rank's flags of step
//      t + 1 to finish step t + 1), so buffer t & 1 is never rewritten while step t is still being read.  If ANY rank's
//      health bit is set, every rank skips the update (the collective form of skip_flag).  No extra launch, no RCCL call.
constexpr int P2P_MAXW = 8;
struct P2PArgs {
    int world, rank, spin_limit, pad;
    float* pub[P2P_MAXW];          // [2][n_pad] floats per rank (this process's mappings; pub[rank] is local memory)
    int32_t* flags[P2P_MAXW];      // [world][gridDim.x] per rank
    int64_t* epoch;                // local: one private step counter per workgroup (as state[4 + b])
    int32_t* health;               // local health word (raised on a poll timeout)
    int64_t n_pad;
};
typedef float p2p_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sys_x4(float* p, p2p_f4 v) {
    // (s_nop: a store of more than 64 bits needs wait states before its data registers may be overwritten, and the compiler's
    //  hazard recognizer does not look inside an asm statement -- without them a v_cndmask scheduled right behind the store
    //  replaced the last dword of lanes 12-15 of every 16: finding 44)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ p2p_f4 ld_sys_x4(const float* p) {
    p2p_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// torch.optim.Adam / AdamW update (torch/optim/adam.py single-tensor path):
//   g += wd*p (Adam)  |  p *= 1 - lr*wd (AdamW)
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= lr/(1-b1^t) * m / ( sqrt(v)/sqrt(1-b2^t) + eps )
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, int decoupled,
                                                   float grad_scale, float clip_norm,
                                                   const float* __restrict__ gnorm, int64_t* state,
                                                   unsigned short* __restrict__ shadow, const ShadowTab tab,
                                                   const int32_t* __restrict__ skip_flag, const P2PArgs p2p) {
    __shared__ int s_p2p_skip;
    // a producer of this step's gradients (the DAG-ERC recurrence kernels) flagged an exchange timeout: the
    // gradients are invalid -- leave parameters, moments and the step counter untouched (checked on the device, no sync)
    const int local_skip = (skip_flag && __hip_atomic_load(skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1 : 0;
    if (p2p.world <= 1 && local_skip) return;
    // first quad of this thread: requested before the (dependent, transcendental) bias-correction math
    const int64_t nq = n >> 2;
    const int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t q0c = q0 < nq ? q0 : 0;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), mv = pv, vv = pv, gv = pv;
    if (nq > 0) {  // uniform
        pv = reinterpret_cast<float4*>(p)[q0c], mv = reinterpret_cast<float4*>(m)[q0c], vv = reinterpret_cast<float4*>(v)[q0c];
        gv = reinterpret_cast<const float4*>(g)[q0c];
    }
    if (p2p.world > 1) {   // (uniform) the gradient exchange, see P2PArgs; single pass: gridDim.x * 256 >= nq (host contract)
        const int b = blockIdx.x, nblk = gridDim.x;
        int64_t* const my_ep = p2p.epoch + b;
        const int ep = (int)(*my_ep + 1);
        const int64_t poff = (int64_t)(ep & 1) * p2p.n_pad + 4 * q0c;
        if (q0 < nq) st_sys_x4(p2p.pub[p2p.rank] + poff, (p2p_f4){gv.x, gv.y, gv.z, gv.w});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if ((int)threadIdx.x < p2p.world) {
            int32_t* f = nullptr;      // flags array of rank threadIdx.x (selected without indexing the by-value struct dynamically)
#pragma unroll
            for (int r = 0; r < P2P_MAXW; ++r)
                if (r == (int)threadIdx.x) f = p2p.flags[r];
            __hip_atomic_store(f + p2p.rank * nblk + b, ep * 2 + local_skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // wait for the same chunk of rank threadIdx.x in the LOCAL flag array
            int32_t* mine = nullptr;
#pragma unroll
            for (int r = 0; r < P2P_MAXW; ++r)
                if (r == p2p.rank) mine = p2p.flags[r];
            const int32_t* w = mine + (int)threadIdx.x * nblk + b;
            int val = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), spins = 0;
            while ((val >> 1) - ep < 0) {
                if (++spins > p2p.spin_limit) {
                    __hip_atomic_store(p2p.health, ERC_HEALTH_RAISED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    val = 2 * ep + 1;      // give up: THIS CHUNK of this rank skips.  The rank's other chunks, and the peers' copies of this
                                           // chunk, may still update: replicas can diverge -- a timeout is fatal for the run
                                           // (FlatParams.check_health raises, checkpoint.save refuses); it is NOT a clean skipped step
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                val = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const unsigned long long any = __ballot((val & 1) != 0);
            if (threadIdx.x == 0) s_p2p_skip = any != 0ull;
        }
        __syncthreads();
        if (threadIdx.x == 0) *my_ep = ep;
        if (s_p2p_skip) return;        // some rank's gradients are invalid: every rank leaves this step out
        p2p_f4 acc = {0.f, 0.f, 0.f, 0.f};
        p2p_f4 part[P2P_MAXW];
#pragma unroll
        for (int r = 0; r < P2P_MAXW; ++r) part[r] = r < p2p.world ? ld_sys_x4(p2p.pub[r] + poff) : (p2p_f4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(part[0]), "+v"(part[1]), "+v"(part[2]), "+v"(part[3]), "+v"(part[4]), "+v"(part[5]), "+v"(part[6]), "+v"(part[7])
                     :
                     : "memory");
#pragma unroll
        for (int r = 0; r < P2P_MAXW; ++r)
            if (r < p2p.world) acc += part[r];
        gv = make_float4(acc.x, acc.y, acc.z, acc.w);
    }
    // Every workgroup keeps its OWN copy of the step count (state[4 + blockIdx.x], all equal between launches): nothing that
    // another workgroup writes is read here, so no arrival counter is needed.  ("The last arriver bumps state[0]" was 274
    // atomics on one word at the tail of the launch: 2.3 us of 10.)
    int64_t* my_step = state + 4 + blockIdx.x;
    const int64_t step = *my_step + 1;
    float gs = grad_scale;
    if (clip_norm > 0.f) {
        const float coef = clip_norm / (gnorm[0] + 1e-6f);
        if (coef < 1.f) gs *= coef;
    }
    AdamCoef ac;
    ac.init(lr, b1, b2, eps, wd, decoupled, gs, step);
    auto update = [&](float& pi, float gi, float& mi, float& vi) { ac.upd(pi, gi, mi, vi); };
    const bool quad_ok = tab.flags & 1;          // host: every range starts on a quad, n0 % 4 == 0, n_el % 4 == 0
    auto to_shadow = [&](int64_t i, float pn) {  // bf16 copies for the bf16 matrix-core products
        if (shadow) shadow_store(shadow, tab, i, pn);
    };
    // 16-byte accesses: one quad per thread and grid stride (the launch sizes the grid for a single pass)
    for (int64_t q = q0; q < nq; q += (int64_t)gridDim.x * 256) {
        if (q != q0) {
            pv = reinterpret_cast<float4*>(p)[q], mv = reinterpret_cast<float4*>(m)[q], vv = reinterpret_cast<float4*>(v)[q];
            gv = reinterpret_cast<const float4*>(g)[q];
        }
        update(pv.x, gv.x, mv.x, vv.x), update(pv.y, gv.y, mv.y, vv.y);
        update(pv.z, gv.z, mv.z, vv.z), update(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4*>(m)[q] = mv, reinterpret_cast<float4*>(v)[q] = vv, reinterpret_cast<float4*>(p)[q] = pv;
        if (shadow) {
            if (quad_ok) shadow_store4(shadow, tab, 4 * q, pv);
            else to_shadow(4 * q, pv.x), to_shadow(4 * q + 1, pv.y), to_shadow(4 * q + 2, pv.z), to_shadow(4 * q + 3, pv.w);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nq << 2) + threadIdx.x;
        float pi = p[i], mi = m[i], vi = v[i];
        update(pi, g[i], mi, vi);
        m[i] = mi, v[i] = vi, p[i] = pi;
        to_shadow(i, pi);
    }
    if (threadIdx.x == 0) {
        *my_step = step;
        if (blockIdx.x == 0) {       // the copies other kernels / the host read; nobody reads them during this launch
            state[0] = step;  // optimizer step
            state[1] += 1;    // RNG offset: a fresh dropout mask next step
        }
    }
    // the private copies no workgroup of THIS launch owns (a launch with another grid -- the weight-gradient launch with the
    // optimizer fused in, csrc/wgrad_bf16.hip -- may own them next time)
    if (blockIdx.x == 0)
        for (int t = gridDim.x + threadIdx.x; t < 512; t += 256) state[4 + t] = step;
}

// Start of a training step: a health word still raised from the previous step (its update was skipped) is counted as an
// event and cleared, so that ONE timeout costs one step, not the rest of the epoch.  Stream order puts this behind the
// previous step's optimizer launch (the only reader of the word) and in front of every kernel that may raise it again.
__global__ void health_roll_kernel(int32_t* live, int32_t* events) {
    if (*live != 0) {
        events[0] += 1;
        *live = 0;
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float scale,
                                                    double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = (double)(g[i] * scale);
        acc += x * x;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void norm_final_kernel(const double* __restrict__ partial, int np,
                                                         float* __restrict__ gnorm) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) gnorm[0] = (float)sqrt(sh[0] + sh[1] + sh[2] + sh[3]);
}

// Diagnostic only (bench.py --clock_probe): one wavefront runs a dependent MFMA chain and records shader cycles
// (s_memtime) against the 100 MHz real-time counter, i.e. the clock the chip holds at that point of the stream.
typedef float probe_f4 __attribute__((ext_vector_type(4)));
__global__ void clock_probe_kernel(unsigned long long* out, int iters) {
    probe_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const float a = (float)threadIdx.x, b = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    if (acc[0] == 12345.f) out[2] = 1;
}

}  // namespace

extern "C" int erc_clock_probe(uint64_t* out, int iters, void* stream) {
    ERC_REQUIRE(out && iters > 0, "clock_probe: bad arguments");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, iters);
    ERC_LAUNCH_CHECK("clock_probe");
    return ERC_OK;
}

// shadow_numel: bf16 elements behind shadow_base.  Strides are non-negative, so the largest (n, k) a range reaches is at
// the largest digits; its destination must lie inside the buffer (a bad table would otherwise write out of bounds).
static int check_shadow_tab(const ShadowTab& tab, int64_t n, int64_t shadow_numel, const char* who) {
    ERC_REQUIRE(tab.n >= 0 && tab.n <= SHADOW_MAX, "%s: %d shadow descriptors (max %d)", who, tab.n, SHADOW_MAX);
    for (int t = 0; t < tab.n; ++t) {
        const ShadowDesc& d = tab.d[t];
        ERC_REQUIRE(d.src_off >= 0 && d.n_el > 0 && d.src_off + d.n_el <= n && d.n_el < (1ll << 31) && d.n0 > 0 && d.n1 > 0 &&
                        d.dst_off >= 0 && d.ld > 0 && (d.mode == 0 || d.mode == 1) && d.sn0 >= 0 && d.sn1 >= 0 && d.sn2 >= 0 &&
                        d.sk0 >= 0 && d.sk1 >= 0 && d.sk2 >= 0 && d.terms >= 1 && d.terms <= 3 && (d.terms == 1 || d.plane_stride > 0),
                    "%s: shadow descriptor %d out of range", who, t);
        const int64_t m0 = (d.n0 < d.n_el ? d.n0 : d.n_el) - 1, q1 = (d.n_el - 1) / d.n0, m1 = q1 < d.n1 ? q1 : d.n1 - 1,
                      m2 = q1 / d.n1;
        const int64_t nmax = m0 * d.sn0 + m1 * d.sn1 + m2 * d.sn2, kmax = m0 * d.sk0 + m1 * d.sk1 + m2 * d.sk2;
        const int64_t last = (d.mode == 0 ? d.dst_off + nmax * d.ld + kmax
                                          : d.dst_off + (((nmax >> 4) * d.ld + (kmax >> 5)) << 9) + 511) + (d.terms - 1) * d.plane_stride;
        ERC_REQUIRE((d.mode == 0 || (kmax >> 5) < d.ld) && last < shadow_numel,
                    "%s: shadow descriptor %d reaches element %lld of a %lld-element shadow buffer", who, t, (long long)last,
                    (long long)shadow_numel);
    }
    return ERC_OK;
}

static int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int decoupled, float grad_scale, float clip_norm, const float* gnorm,
                       int64_t* state, void* shadow_base, int64_t shadow_numel, const ShadowTab& tab_in,
                       const int32_t* skip_flag, void* stream, const P2PArgs* p2p_in = nullptr) {
    ShadowTab tab = tab_in;
    P2PArgs p2p{};
    if (p2p_in) p2p = *p2p_in;
    // quad fast path (flags bit 0): every range starts on a quad of the flat buffer and has rows of a multiple of 4
    // elements; bit 1 + t: range t runs along k with k % 4 == 0 for every quad and 8-byte aligned destinations
    tab.flags = tab.n > 0;
    for (int t = 0; t < tab.n; ++t) {
        const ShadowDesc& d = tab.d[t];
        if (d.src_off % 4 || d.n0 % 4 || d.n_el % 4) tab.flags &= ~1;
        const bool kq = d.sn0 == 0 && d.sk0 == 1 && d.sk1 % 4 == 0 && d.sk2 % 4 == 0 && d.dst_off % 4 == 0 && d.plane_stride % 4 == 0 &&
                        (d.mode == 1 || d.ld % 4 == 0) && ((uintptr_t)shadow_base & 7) == 0;
        if (kq) tab.flags |= 2 << t;
    }
    ERC_REQUIRE(p && g && m && v && state && n > 0, "adam_step: bad arguments");
    ERC_REQUIRE(clip_norm <= 0.f || gnorm, "adam_step: clip_norm needs gnorm");
    ERC_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: 16-byte alignment");
    if (int rc = check_shadow_tab(tab, n, shadow_numel, "adam_step")) return rc;
    int grid = (int)((n / 4 + 255) / 256);   // one float4 per thread
    if (grid < 1) grid = 1;
    if (grid > 512) grid = 512;  // one arrival atomic per block on a single word: keep the count low
    ERC_REQUIRE(p2p.world <= 1 || ((int64_t)grid * 256 >= n / 4 && (n & 3) == 0 && clip_norm <= 0.f),
                "adam_step_p2p: the fused exchange takes n <= 524288 parameters, n %% 4 == 0, no clip-norm (use the RCCL all-reduce)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       decoupled, grad_scale, clip_norm, gnorm, state, tab.n > 0 ? (unsigned short*)shadow_base : nullptr, tab,
                       skip_flag, p2p);
    ERC_LAUNCH_CHECK("adam_step");
    return ERC_OK;
}

extern "C" int erc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                             float clip_norm, const float* gnorm, int64_t* state, void* bf16_shadow,
                             int64_t shadow_off, int64_t shadow_n, const int32_t* skip_flag, void* stream) {
    ShadowTab tab{};
    if (bf16_shadow && shadow_n > 0) {
        tab.n = 1;
        tab.d[0] = ShadowDesc{shadow_off, shadow_n, 0, (int32_t)shadow_n, 1, 0, 0, 0, 1, 0, 0, (int32_t)shadow_n, 0, 0, 1, 0};   // identity: k = idx
    }
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale, clip_norm, gnorm, state,
                       bf16_shadow, shadow_n, tab, skip_flag, stream);
}

// Same step with a table of bf16 shadow ranges (ErcShadowTab in ercgraft.h, a HOST struct passed by value to the kernel).
extern "C" int erc_adam_step_tab(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                                 float clip_norm, const float* gnorm, int64_t* state, void* shadow_base,
                                 int64_t shadow_numel, const ErcShadowTab* tab_host, const int32_t* skip_flag, void* stream) {
    static_assert(sizeof(ShadowTab) == sizeof(ErcShadowTab), "shadow table layout");
    ShadowTab tab{};
    if (tab_host) memcpy(&tab, tab_host, sizeof(tab));
    ERC_REQUIRE(tab.n == 0 || shadow_base, "adam_step_tab: shadow table without a shadow buffer");
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale, clip_norm, gnorm, state,
                       shadow_base, shadow_numel, tab, skip_flag, stream);
}

// (Re)build every shadow range of the table from the fp32 parameters (after loading a state dict, or when no optimizer
// maintains them).
extern "C" int erc_shadow_refresh(const float* p, int64_t n, void* shadow_base, int64_t shadow_numel,
                                  const ErcShadowTab* tab_host, void* stream) {
    ERC_REQUIRE(p && shadow_base && tab_host, "shadow_refresh: null pointer");
    ShadowTab tab{};
    memcpy(&tab, tab_host, sizeof(tab));
    ERC_REQUIRE(tab.n > 0, "shadow_refresh: empty table");
    if (int rc = check_shadow_tab(tab, n, shadow_numel, "shadow_refresh")) return rc;
    hipLaunchKernelGGL(shadow_refresh_kernel, dim3(64, tab.n), dim3(256), 0, (hipStream_t)stream, p, (unsigned short*)shadow_base, tab);
    ERC_LAUNCH_CHECK("shadow_refresh");
    return ERC_OK;
}

extern "C" int erc_health_roll(int32_t* health, int32_t* events, void* stream) {
    ERC_REQUIRE(health && events, "health_roll: null pointer");
    hipLaunchKernelGGL(health_roll_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, health, events);
    ERC_LAUNCH_CHECK("health_roll");
    return ERC_OK;
}

extern "C" int erc_grad_norm(const float* g, int64_t n, float grad_scale, float* gnorm, float* ws, void* stream) {
    ERC_REQUIRE(g && gnorm && ws && n > 0, "grad_norm: bad arguments");
    ERC_REQUIRE(((uintptr_t)ws & 7) == 0, "grad_norm: ws must be 8-byte aligned");
    int grid = (int)((n + 255) / 256);
    if (grid > 512) grid = 512;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, st, g, n, grad_scale, (double*)ws);
    ERC_LAUNCH_CHECK("grad_norm.sumsq");
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, grid, gnorm);
    ERC_LAUNCH_CHECK("grad_norm.final");
    return ERC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Peer-mapped buffers of the fused gradient exchange (P2PArgs above).  erc_p2p_alloc: device memory other processes can
// map (uncached where the runtime offers it: the flag words and publish buffers are written and read with system-scope
// accesses only) + its 64-byte IPC handle; erc_p2p_open maps a peer's handle into this process.
extern "C" int erc_p2p_alloc(int64_t bytes, void** ptr, void* handle64) {
    ERC_REQUIRE(bytes > 0 && ptr && handle64, "p2p_alloc: bad arguments");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    void* q = nullptr;
    // UNCACHED (fine-grained) memory or nothing: the exchange relies on peers seeing sc0 sc1 stores made while the kernel runs;
    // plain coarse-grained hipMalloc memory does not promise that across devices (stale sums, bounded-wait timeouts).  The
    // caller falls back to the RCCL all-reduce when this fails (engine.FusedAdam.enable_p2p).
    if (hipExtMallocWithFlags(&q, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        erc_set_error("p2p_alloc: the runtime refused %lld bytes of uncached device memory (no coarse-grained fallback: peers must see "
                      "stores of a running kernel)", (long long)bytes);
        return ERC_E_ARG;
    }
    ERC_REQUIRE(hipMemset(q, 0, (size_t)bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess, "p2p_alloc: memset failed");
    hipIpcMemHandle_t h;
    ERC_REQUIRE(hipIpcGetMemHandle(&h, q) == hipSuccess, "p2p_alloc: hipIpcGetMemHandle failed: %s", hipGetErrorString(hipGetLastError()));
    memcpy(handle64, &h, 64);
    *ptr = q;
    return ERC_OK;
}
extern "C" int erc_p2p_open(const void* handle64, void** ptr) {
    ERC_REQUIRE(handle64 && ptr, "p2p_open: bad arguments");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    ERC_REQUIRE(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess, "p2p_open: hipIpcOpenMemHandle failed: %s",
                hipGetErrorString(hipGetLastError()));
    return ERC_OK;
}
extern "C" int erc_p2p_close(void* ptr) {
    ERC_REQUIRE(hipIpcCloseMemHandle(ptr) == hipSuccess, "p2p_close failed");
    return ERC_OK;
}
extern "C" int erc_p2p_free(void* ptr) {
    ERC_REQUIRE(hipFree(ptr) == hipSuccess, "p2p_free failed");
    return ERC_OK;
}

// erc_adam_step_tab with the gradient exchange fused in (ErcP2P in ercgraft.h): g is this rank's LOCAL gradient, the update
// uses the rank-ordered sum over all ranks times grad_scale; skip_flag must be p2p->health.
extern "C" int erc_adam_step_p2p(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int decoupled, float grad_scale, int64_t* state,
                                 void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host, const ErcP2P* x,
                                 void* stream) {
    ERC_REQUIRE(x && x->world >= 2 && x->world <= P2P_MAXW && x->rank >= 0 && x->rank < x->world && x->epoch && x->health &&
                    x->n_pad >= n && x->n_pad % 4 == 0, "adam_step_p2p: bad exchange descriptor (world <= %d)", P2P_MAXW);
    ShadowTab tab{};
    if (tab_host) memcpy(&tab, tab_host, sizeof(tab));
    ERC_REQUIRE(tab.n == 0 || shadow_base, "adam_step_p2p: shadow table without a shadow buffer");
    P2PArgs a{};
    a.world = x->world, a.rank = x->rank, a.spin_limit = x->spin_limit > 0 ? x->spin_limit : 4000000;
    for (int r = 0; r < x->world; ++r) {
        ERC_REQUIRE(x->pub[r] && x->flags[r] && ((uintptr_t)x->pub[r] & 15) == 0, "adam_step_p2p: peer %d not mapped", r);
        a.pub[r] = (float*)x->pub[r], a.flags[r] = (int32_t*)x->flags[r];
    }
    a.epoch = (int64_t*)x->epoch, a.health = (int32_t*)x->health, a.n_pad = x->n_pad;
    return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale, 0.f, nullptr, state, shadow_base,
                       shadow_numel, tab, (const int32_t*)x->health, stream, &a);
}
