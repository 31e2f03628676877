// K5 (BatchNorm1d + LeakyReLU over the nodes of this rank) and S3 (cross entropy).
//
// BatchNorm: column statistics of an [N,F] matrix are accumulated in fp64 by
// BN_G workgroups into per-workgroup partials; every workgroup of the apply
// kernel re-reduces the (tiny) partial table, so there is no third launch, no
// atomics, and the result does not depend on scheduling.
#include "erc_common.h"

namespace {

constexpr int BN_G = 64;  // workgroups of the statistics pass

// partial[g][0..F) = sum_rows a(row,c) ; partial[g][F..2F) = sum_rows b(row,c)
// MODE 0: a = x, b = x*x.  MODE 1 (backward): a = dz, b = dz*xhat.
template <int MODE>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int ldx, int N, int F,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ saved, float slope,
                                                       const float* __restrict__ dy, int lddy,
                                                       double* __restrict__ partial) {
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    __shared__ double sh[2][128];
    double a = 0.0, b = 0.0;
    if (c < F) {
        float mean = 0.f, rstd = 0.f, ga = 0.f, be = 0.f;
        if (MODE == 1) mean = saved[c], rstd = saved[F + c], ga = gamma[c], be = beta[c];
        for (int row = blockIdx.x * 2 + half; row < N; row += 2 * BN_G) {
            const float v = x[(int64_t)row * ldx + c];
            if (MODE == 0) {
                a += (double)v;
                b += (double)v * (double)v;
            } else {
                const float xh = (v - mean) * rstd;
                const float zz = xh * ga + be;
                const float dz = dy[(int64_t)row * lddy + c] * (zz > 0.f ? 1.f : slope);
                a += (double)dz;
                b += (double)dz * (double)xh;
            }
        }
    }
    if (half == 1) sh[0][c] = a, sh[1][c] = b;
    __syncthreads();
    if (half == 0 && c < F) {
        partial[(int64_t)blockIdx.x * 2 * F + c] = a + sh[0][c];
        partial[(int64_t)blockIdx.x * 2 * F + F + c] = b + sh[1][c];
    }
}

__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, int ldx, int N, int F,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* running_mean,
                                                           float* running_var, float momentum, float eps, float slope,
                                                           int training, float* __restrict__ saved,
                                                           float* __restrict__ y, int ldy,
                                                           const double* __restrict__ partial) {
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    __shared__ float s_mean[128], s_rstd[128];
    if (half == 0 && c < F) {
        float mean, rstd;
        if (training) {
            double s = 0.0, ss = 0.0;
            for (int g = 0; g < BN_G; ++g) {
                s += partial[(int64_t)g * 2 * F + c];
                ss += partial[(int64_t)g * 2 * F + F + c];
            }
            const double m = s / (double)N;
            double var = ss / (double)N - m * m;
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            rstd = (float)(1.0 / sqrt(var + (double)eps));
            if (blockIdx.x == 0) {
                saved[c] = mean;
                saved[F + c] = rstd;
                const double unbiased = N > 1 ? var * (double)N / (double)(N - 1) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        } else {
            mean = running_mean[c];
            rstd = 1.0f / sqrtf(running_var[c] + eps);
        }
        s_mean[c] = mean;
        s_rstd[c] = rstd;
    }
    __syncthreads();
    if (c >= F) return;
    const float mean = s_mean[c], rstd = s_rstd[c], ga = gamma[c], be = beta[c];
    for (int row = blockIdx.x * 2 + half; row < N; row += 2 * gridDim.x) {
        const float z = (x[(int64_t)row * ldx + c] - mean) * rstd * ga + be;
        y[(int64_t)row * ldy + c] = z > 0.f ? z : z * slope;
    }
}

__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float* __restrict__ x, int ldx, int N, int F,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ saved, float slope,
                                                           const float* __restrict__ dy, int lddy,
                                                           float* __restrict__ dx, int lddx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           const double* __restrict__ partial) {
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    __shared__ float s_a[128], s_b[128];
    if (half == 0 && c < F) {
        double s = 0.0, ss = 0.0;
        for (int g = 0; g < BN_G; ++g) {
            s += partial[(int64_t)g * 2 * F + c];
            ss += partial[(int64_t)g * 2 * F + F + c];
        }
        s_a[c] = (float)(s / (double)N);   // mean of dz
        s_b[c] = (float)(ss / (double)N);  // mean of dz*xhat
        if (blockIdx.x == 0) {
            dbeta[c] = (float)s;
            dgamma[c] = (float)ss;
        }
    }
    __syncthreads();
    if (c >= F) return;
    const float mean = saved[c], rstd = saved[F + c], ga = gamma[c], be = beta[c];
    const float ma = s_a[c], mb = s_b[c];
    for (int row = blockIdx.x * 2 + half; row < N; row += 2 * gridDim.x) {
        const float xh = (x[(int64_t)row * ldx + c] - mean) * rstd;
        const float zz = xh * ga + be;
        const float dz = dy[(int64_t)row * lddy + c] * (zz > 0.f ? 1.f : slope);
        dx[(int64_t)row * lddx + c] = ga * rstd * (dz - ma - xh * mb);
    }
}

// ------------------------------------------------------------ cross entropy
// Rows are spread over up to CE_MAXWG workgroups (the gradient needs only the normaliser, which every workgroup
// recomputes from the labels); loss / accuracy partials are combined by the LAST workgroup to finish, in
// workgroup order, so the result is independent of scheduling.  stats[] layout: [0] loss [1] #correct [2] sum of
// weights [8 + 2w], [9 + 2w] partials of workgroup w, [4] (as int) arrival counter (zero between calls).
constexpr int CE_MAXWG = 64;

__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, int ld, int C,
                                                            int n_rows, const int32_t* __restrict__ row_map,
                                                            const int64_t* __restrict__ labels,
                                                            const float* __restrict__ weight, float grad_scale,
                                                            float* __restrict__ dlogits, int lddl,
                                                            float* __restrict__ stats) {
    __shared__ double red[256];
    double* const s_red = red;
    __shared__ double s_wsum;
    __shared__ int s_last;
    const int tid = threadIdx.x;
    double wacc = 0.0;
    if (weight) {
        for (int i = tid; i < n_rows; i += 256) wacc += (double)weight[labels[i]];
        red[tid] = wacc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) s_wsum = red[0];
    } else if (tid == 0) {
        s_wsum = (double)n_rows;
    }
    __syncthreads();
    const double wsum = s_wsum;
    const float inv_w = (float)(1.0 / wsum);
    double lacc = 0.0;
    int hit = 0;
    for (int i = blockIdx.x * 256 + tid; i < n_rows; i += gridDim.x * 256) {
        const int64_t row = row_map ? (int64_t)row_map[i] : (int64_t)i;
        const float* z = logits + row * ld;
        const int y = (int)labels[i];
        float mx = z[0];
        int am = 0;
        for (int c = 1; c < C; ++c) {
            const float v = z[c];
            if (v > mx) mx = v, am = c;
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
        const float lse = mx + logf(se);
        const float w = weight ? weight[y] : 1.f;
        lacc += (double)(w * (lse - z[y]));
        hit += (am == y) ? 1 : 0;
        if (dlogits) {
            float* d = dlogits + row * lddl;
            const float coef = w * inv_w * grad_scale;
            for (int c = 0; c < C; ++c) d[c] = coef * (expf(z[c] - lse) - (c == y ? 1.f : 0.f));
        }
    }
    __syncthreads();
    red[tid] = lacc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double lsum = red[0];
    __syncthreads();
    red[tid] = (double)hit;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        stats[8 + 2 * blockIdx.x] = (float)lsum;
        stats[9 + 2 * blockIdx.x] = (float)red[0];
        __threadfence();
        const int prev = atomicAdd(reinterpret_cast<int*>(stats + 4), 1);
        s_last = (prev == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (s_last) {  // the last workgroup combines the partials: one (parallel) load per thread, fixed-order tree
        __threadfence();
        double l = 0.0, h = 0.0;
        if (tid < (int)gridDim.x) {
            l = (double)__hip_atomic_load(stats + 8 + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h = (double)__hip_atomic_load(stats + 9 + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        s_red[tid] = l;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        const double lsum = s_red[0];
        __syncthreads();
        s_red[tid] = h;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        if (tid == 0) {
            stats[0] = (float)(lsum / wsum);
            stats[1] = (float)s_red[0];
            stats[2] = (float)wsum;
            *reinterpret_cast<int*>(stats + 4) = 0;
        }
    }
}

// ------------------------------------------------------------ fused classifier tail
// logits = Z W^T + b (C <= 8 classes, F <= 128 features), cross entropy, and the gradient wrt Z through the
// relu/dropout mask -- cls[3] + F.cross_entropy + their backward (track_mm/cogmen.py:116-122,185) in ONE launch
// instead of GEMM + CE + GEMM.  One wavefront per row, lane l owns features l and l+64.
constexpr int HC_MAXC = 8;
constexpr int HC_MAXWG = 256;  // workgroups (one arrival atomic each)

__global__ __launch_bounds__(256) void head_ce_kernel(const float* __restrict__ Z, int ldz, int F, int C, int n_rows,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      const int64_t* __restrict__ labels,
                                                      const float* __restrict__ weight, float mask_scale,
                                                      float* __restrict__ logits, int ldl, float* __restrict__ dlogits,
                                                      int lddl, float* __restrict__ dZ, int lddz,
                                                      float* __restrict__ stats, int rows_per_wg) {
    __shared__ double s_red[256];
    __shared__ double s_wsum;
    __shared__ float s_part[4][2];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (weight) {
        double wacc = 0.0;
        for (int i = tid; i < n_rows; i += 256) wacc += (double)weight[labels[i]];
        s_red[tid] = wacc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        if (tid == 0) s_wsum = s_red[0];
    } else if (tid == 0) {
        s_wsum = (double)n_rows;
    }
    __syncthreads();
    const double wsum = s_wsum;
    const float inv_w = (float)(1.0 / wsum);
    const bool h0 = lane < F, h1 = lane + 64 < F;
    const int c0 = min(lane, F - 1), c1 = min(lane + 64, F - 1);
    // unconditional clamped loads, masked afterwards (a guarded load costs a full round trip each)
    float w0[HC_MAXC], w1[HC_MAXC], bk[HC_MAXC];
    const float m0 = h0 ? 1.f : 0.f, m1 = h1 ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < HC_MAXC; ++k) {
        const int kc = min(k, C - 1);
        w0[k] = W[(int64_t)kc * F + c0] * m0;
        w1[k] = W[(int64_t)kc * F + c1] * m1;
        bk[k] = bias[kc];
    }
    float lacc = 0.f, hits = 0.f;
    for (int rr = wave; rr < rows_per_wg; rr += 4) {
        const int row = blockIdx.x * rows_per_wg + rr;
        if (row >= n_rows) break;
        const float z0 = Z[(int64_t)row * ldz + c0], z1 = Z[(int64_t)row * ldz + c1];
        const int y = (int)labels[row];
        float lg[HC_MAXC];
#pragma unroll
        for (int k = 0; k < HC_MAXC; ++k) lg[k] = wave_sum(z0 * w0[k] + z1 * w1[k]) + bk[k];
        float mx = lg[0];
        int am = 0;
#pragma unroll
        for (int k = 1; k < HC_MAXC; ++k)
            if (k < C && lg[k] > mx) mx = lg[k], am = k;
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < HC_MAXC; ++k)
            if (k < C) se += expf(lg[k] - mx);
        const float lse = mx + logf(se);
        float wy = 1.f;
        if (weight) wy = weight[y];   // wave-uniform branch
        float ly = 0.f, g0 = 0.f, g1 = 0.f;
        const float coef = wy * inv_w;
#pragma unroll
        for (int k = 0; k < HC_MAXC; ++k) {
            if (k < C) {
                const float d = coef * (expf(lg[k] - lse) - (k == y ? 1.f : 0.f));
                if (k == y) ly = lg[k];
                g0 += d * w0[k];
                g1 += d * w1[k];
                if (lane == k) {
                    logits[(int64_t)row * ldl + k] = lg[k];
                    dlogits[(int64_t)row * lddl + k] = d;
                }
            }
        }
        if (h0) dZ[(int64_t)row * lddz + lane] = z0 > 0.f ? g0 * mask_scale : 0.f;
        if (h1) dZ[(int64_t)row * lddz + lane + 64] = z1 > 0.f ? g1 * mask_scale : 0.f;
        lacc += wy * (lse - ly);
        hits += (am == y) ? 1.f : 0.f;
    }
    if (lane == 0) s_part[wave][0] = lacc, s_part[wave][1] = hits;
    __syncthreads();
    if (tid == 0) {
        stats[8 + 2 * blockIdx.x] = s_part[0][0] + s_part[1][0] + s_part[2][0] + s_part[3][0];
        stats[9 + 2 * blockIdx.x] = s_part[0][1] + s_part[1][1] + s_part[2][1] + s_part[3][1];
        __threadfence();
        const int prev = atomicAdd(reinterpret_cast<int*>(stats + 4), 1);
        s_last = (prev == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (s_last) {  // the last workgroup combines the partials: one (parallel) load per thread, fixed-order tree
        __threadfence();
        double l = 0.0, h = 0.0;
        if (tid < (int)gridDim.x) {
            l = (double)__hip_atomic_load(stats + 8 + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h = (double)__hip_atomic_load(stats + 9 + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        s_red[tid] = l;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        const double lsum = s_red[0];
        __syncthreads();
        s_red[tid] = h;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        if (tid == 0) {
            stats[0] = (float)(lsum / wsum);
            stats[1] = (float)s_red[0];
            stats[2] = (float)wsum;
            *reinterpret_cast<int*>(stats + 4) = 0;
        }
    }
}

}  // namespace

extern "C" int64_t erc_head_ce_stats_floats(int n_rows) { return 16 + 2 * HC_MAXWG; }

extern "C" int erc_head_ce(const float* Z, int ldz, int F, int C, int n_rows, const float* W, const float* bias,
                           const int64_t* labels, const float* weight, float mask_scale, float* logits, int ldl,
                           float* dlogits, int lddl, float* dZ, int lddz, float* stats, void* stream) {
    ERC_REQUIRE(Z && W && bias && labels && logits && dlogits && dZ && stats, "head_ce: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && C > 0 && C <= HC_MAXC && n_rows > 0, "head_ce: F=%d C=%d n_rows=%d unsupported", F, C, n_rows);
    int grid = erc_cdiv(n_rows, 16);
    if (grid > HC_MAXWG) grid = HC_MAXWG;
    const int rows_per_wg = erc_cdiv(n_rows, grid);
    grid = erc_cdiv(n_rows, rows_per_wg);
    hipLaunchKernelGGL(head_ce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Z, ldz, F, C, n_rows, W, bias,
                       labels, weight, mask_scale, logits, ldl, dlogits, lddl, dZ, lddz, stats, rows_per_wg);
    ERC_LAUNCH_CHECK("head_ce");
    return ERC_OK;
}

extern "C" int64_t erc_bn_ws_floats(int F) { return (int64_t)BN_G * 2 * F * 2 + 16; }

extern "C" int erc_bn_lrelu_fwd(const float* x, int ldx, int N, int F, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float slope,
                                int training, float* saved, float* y, int ldy, float* ws, void* stream) {
    ERC_REQUIRE(x && gamma && beta && running_mean && running_var && y && ws, "bn_lrelu_fwd: null pointer");
    ERC_REQUIRE(!training || saved, "bn_lrelu_fwd: training needs saved[2F]");
    ERC_REQUIRE(N > 0 && F > 0 && F <= 128, "bn_lrelu_fwd: N=%d F=%d unsupported", N, F);
    ERC_REQUIRE(((uintptr_t)ws & 7) == 0, "bn_lrelu_fwd: ws must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)ws;
    if (training) {
        hipLaunchKernelGGL(bn_stats_kernel<0>, dim3(BN_G), dim3(256), 0, st, x, ldx, N, F, gamma, beta, saved, slope,
                           (const float*)nullptr, 0, partial);
        ERC_LAUNCH_CHECK("bn_stats");
    }
    int grid = erc_cdiv(N, 8);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(grid), dim3(256), 0, st, x, ldx, N, F, gamma, beta, running_mean,
                       running_var, momentum, eps, slope, training, saved, y, ldy, partial);
    ERC_LAUNCH_CHECK("bn_apply_fwd");
    return ERC_OK;
}

extern "C" int erc_bn_lrelu_bwd(const float* x, int ldx, int N, int F, const float* gamma, const float* beta,
                                const float* saved, float slope, const float* dy, int lddy, float* dx, int lddx,
                                float* dgamma, float* dbeta, float* ws, void* stream) {
    ERC_REQUIRE(x && gamma && beta && saved && dy && dx && dgamma && dbeta && ws, "bn_lrelu_bwd: null pointer");
    ERC_REQUIRE(N > 0 && F > 0 && F <= 128, "bn_lrelu_bwd: N=%d F=%d unsupported", N, F);
    ERC_REQUIRE(((uintptr_t)ws & 7) == 0, "bn_lrelu_bwd: ws must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)ws;
    hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(BN_G), dim3(256), 0, st, x, ldx, N, F, gamma, beta, saved, slope, dy,
                       lddy, partial);
    ERC_LAUNCH_CHECK("bn_bwd_stats");
    int grid = erc_cdiv(N, 8);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(grid), dim3(256), 0, st, x, ldx, N, F, gamma, beta, saved, slope, dy,
                       lddy, dx, lddx, dgamma, dbeta, partial);
    ERC_LAUNCH_CHECK("bn_apply_bwd");
    return ERC_OK;
}

extern "C" int erc_cross_entropy(const float* logits, int ld, int C, int n_rows, const int32_t* row_map,
                                 const int64_t* labels, const float* weight, float grad_scale, float* dlogits,
                                 int lddl, float* stats, void* stream) {
    ERC_REQUIRE(logits && labels && stats, "cross_entropy: null pointer");
    ERC_REQUIRE(C > 0 && n_rows > 0 && ld >= C, "cross_entropy: C=%d n_rows=%d ld=%d", C, n_rows, ld);
    int grid = erc_cdiv(n_rows, 256);
    if (grid > CE_MAXWG) grid = CE_MAXWG;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, ld, C, n_rows,
                       row_map, labels, weight, grad_scale, dlogits, lddl, stats);
    ERC_LAUNCH_CHECK("cross_entropy");
    return ERC_OK;
}
