// K3 / K4: message passing over the dialogue graph, one wavefront per node.
//
// Both operators are written as GATHERS over a CSR (by target for the forward
// and the target-side backward, by source for the source-side backward), so
// every output row is produced by exactly one wavefront in a fixed edge order:
// deterministic, no float atomics.  Feature rows are F <= 128 floats; lane l
// owns channels l and l+64.
#include "erc_common.h"

namespace {

constexpr int MAX_R = 8;

// ---------------------------------------------------------------- RGCN mean
__global__ __launch_bounds__(256) void rgcn_mean_fwd_kernel(const float* __restrict__ x, int ldx, int F, int R, int N,
                                                            const int32_t* __restrict__ in_ptr,
                                                            const int32_t* __restrict__ in_src,
                                                            const int32_t* __restrict__ in_typ,
                                                            float* __restrict__ Mout, int ldm,
                                                            float* __restrict__ inv_cnt) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int c0 = lane, c1 = lane + 64;
    const bool h0 = c0 < F, h1 = c1 < F;
    float s0[MAX_R], s1[MAX_R];
    int cnt[MAX_R];
#pragma unroll
    for (int r = 0; r < MAX_R; ++r) s0[r] = s1[r] = 0.f, cnt[r] = 0;
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    for (int e = e0; e < e1; ++e) {
        const int j = in_src[e];
        const int t = in_typ[e];
        const float v0 = h0 ? x[(int64_t)j * ldx + c0] : 0.f;
        const float v1 = h1 ? x[(int64_t)j * ldx + c1] : 0.f;
#pragma unroll
        for (int r = 0; r < MAX_R; ++r) {
            const bool hit = (t == r);
            s0[r] += hit ? v0 : 0.f;
            s1[r] += hit ? v1 : 0.f;
            cnt[r] += hit ? 1 : 0;
        }
    }
    float* mrow = Mout + (int64_t)i * ldm;
#pragma unroll
    for (int r = 0; r < MAX_R; ++r) {
        if (r < R) {
            const float inv = cnt[r] > 0 ? 1.0f / (float)cnt[r] : 0.f;
            // mean = sum / count (a division, like scatter-mean) -- inv is only kept for the backward
            if (h0) mrow[r * F + c0] = cnt[r] > 0 ? s0[r] / (float)cnt[r] : 0.f;
            if (h1) mrow[r * F + c1] = cnt[r] > 0 ? s1[r] / (float)cnt[r] : 0.f;
            if (lane == 0) inv_cnt[(int64_t)i * R + r] = inv;
        }
    }
    if (h0) mrow[R * F + c0] = x[(int64_t)i * ldx + c0];
    if (h1) mrow[R * F + c1] = x[(int64_t)i * ldx + c1];
}

__global__ __launch_bounds__(256) void rgcn_mean_bwd_kernel(const float* __restrict__ dM, int ldm, int F, int R, int N,
                                                            const int32_t* __restrict__ out_ptr,
                                                            const int32_t* __restrict__ out_dst,
                                                            const int32_t* __restrict__ out_typ,
                                                            const float* __restrict__ inv_cnt,
                                                            float* __restrict__ dx, int lddx) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    const int c0 = lane, c1 = lane + 64;
    const bool h0 = c0 < F, h1 = c1 < F;
    float a0 = h0 ? dM[(int64_t)j * ldm + R * F + c0] : 0.f;
    float a1 = h1 ? dM[(int64_t)j * ldm + R * F + c1] : 0.f;
    const int e0 = out_ptr[j], e1 = out_ptr[j + 1];
    for (int e = e0; e < e1; ++e) {
        const int i = out_dst[e];
        const int t = out_typ[e];
        if (t >= R) continue;
        const float w = inv_cnt[(int64_t)i * R + t];
        const float* row = dM + (int64_t)i * ldm + t * F;
        if (h0) a0 += row[c0] * w;
        if (h1) a1 += row[c1] * w;
    }
    if (h0) dx[(int64_t)j * lddx + c0] = a0;
    if (h1) dx[(int64_t)j * lddx + c1] = a1;
}

// ------------------------------------------------- TransformerConv, heads=1
__global__ __launch_bounds__(256) void tconv_fwd_kernel(const float* __restrict__ qkvs, int ld, int F, int N,
                                                        float scale, const int32_t* __restrict__ in_ptr,
                                                        const int32_t* __restrict__ in_src, float* __restrict__ out,
                                                        int ldo, float* __restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int c0 = lane, c1 = lane + 64;
    const bool h0 = c0 < F, h1 = c1 < F;
    const float* qi = qkvs + (int64_t)i * ld;
    const float q0 = h0 ? qi[c0] : 0.f, q1 = h1 ? qi[c1] : 0.f;
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    // pass 1: max of the raw scores
    float mx = -INFINITY;
    for (int e = e0; e < e1; ++e) {
        const float* kj = qkvs + (int64_t)in_src[e] * ld + F;
        const float part = (h0 ? q0 * kj[c0] : 0.f) + (h1 ? q1 * kj[c1] : 0.f);
        mx = fmaxf(mx, wave_sum(part) * scale);
    }
    // pass 2: exp, denominator, weighted values (unnormalised); lane (e-e0) keeps exp_e of the first 64 edges
    float den = 0.f, o0 = 0.f, o1 = 0.f, mine = 0.f;
    for (int e = e0; e < e1; ++e) {
        const float* kj = qkvs + (int64_t)in_src[e] * ld + F;
        const float part = (h0 ? q0 * kj[c0] : 0.f) + (h1 ? q1 * kj[c1] : 0.f);
        const float pexp = expf(wave_sum(part) * scale - mx);
        den += pexp;
        const float* vj = qkvs + (int64_t)in_src[e] * ld + 2 * F;
        if (h0) o0 += pexp * vj[c0];
        if (h1) o1 += pexp * vj[c1];
        if (e - e0 == lane) mine = pexp;
    }
    const float inv = 1.0f / (den + 1e-16f);
    const float* si = qi + 3 * F;
    if (h0) out[(int64_t)i * ldo + c0] = o0 * inv + si[c0];
    if (h1) out[(int64_t)i * ldo + c1] = o1 * inv + si[c1];
    // pass 3: normalised weights for the backward; no memory round trip inside the wave
    if (e0 + lane < e1) alpha[e0 + lane] = mine * inv;
    for (int e = e0 + 64; e < e1; ++e) {  // degree > 64 (never for window graphs): recompute
        const float* kj = qkvs + (int64_t)in_src[e] * ld + F;
        const float part = (h0 ? q0 * kj[c0] : 0.f) + (h1 ? q1 * kj[c1] : 0.f);
        const float a = expf(wave_sum(part) * scale - mx) * inv;
        if (lane == 0) alpha[e] = a;
    }
}

__global__ __launch_bounds__(256) void tconv_bwd_target_kernel(const float* __restrict__ qkvs, int ld, int F, int N,
                                                               float scale, const int32_t* __restrict__ in_ptr,
                                                               const int32_t* __restrict__ in_src,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ dout, int lddo,
                                                               float* __restrict__ dqkvs,
                                                               float* __restrict__ dscore) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int c0 = lane, c1 = lane + 64;
    const bool h0 = c0 < F, h1 = c1 < F;
    const float g0 = h0 ? dout[(int64_t)i * lddo + c0] : 0.f;
    const float g1 = h1 ? dout[(int64_t)i * lddo + c1] : 0.f;
    const int e0 = in_ptr[i], e1 = in_ptr[i + 1];
    // t = sum_e alpha_e * (dout_i . v_src(e))
    float t = 0.f;
    for (int e = e0; e < e1; ++e) {
        const float* vj = qkvs + (int64_t)in_src[e] * ld + 2 * F;
        const float da = wave_sum((h0 ? g0 * vj[c0] : 0.f) + (h1 ? g1 * vj[c1] : 0.f));
        t += alpha[e] * da;
    }
    float dq0 = 0.f, dq1 = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int j = in_src[e];
        const float* vj = qkvs + (int64_t)j * ld + 2 * F;
        const float da = wave_sum((h0 ? g0 * vj[c0] : 0.f) + (h1 ? g1 * vj[c1] : 0.f));
        const float ds = alpha[e] * (da - t) * scale;  // dL/d(q.k)
        if (lane == 0) dscore[e] = ds;
        const float* kj = qkvs + (int64_t)j * ld + F;
        if (h0) dq0 += ds * kj[c0];
        if (h1) dq1 += ds * kj[c1];
    }
    float* d = dqkvs + (int64_t)i * ld;
    if (h0) d[c0] = dq0, d[3 * F + c0] = g0;
    if (h1) d[c1] = dq1, d[3 * F + c1] = g1;
}

__global__ __launch_bounds__(256) void tconv_bwd_source_kernel(const float* __restrict__ qkvs, int ld, int F, int N,
                                                               const int32_t* __restrict__ out_ptr,
                                                               const int32_t* __restrict__ out_dst,
                                                               const int32_t* __restrict__ out_eid,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ dscore,
                                                               const float* __restrict__ dout, int lddo,
                                                               float* __restrict__ dqkvs) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    const int c0 = lane, c1 = lane + 64;
    const bool h0 = c0 < F, h1 = c1 < F;
    float dk0 = 0.f, dk1 = 0.f, dv0 = 0.f, dv1 = 0.f;
    const int e0 = out_ptr[j], e1 = out_ptr[j + 1];
    for (int e = e0; e < e1; ++e) {
        const int i = out_dst[e];
        const int id = out_eid[e];
        const float ds = dscore[id], al = alpha[id];
        const float* qi = qkvs + (int64_t)i * ld;
        const float* gi = dout + (int64_t)i * lddo;
        if (h0) dk0 += ds * qi[c0], dv0 += al * gi[c0];
        if (h1) dk1 += ds * qi[c1], dv1 += al * gi[c1];
    }
    float* d = dqkvs + (int64_t)j * ld;
    if (h0) d[F + c0] = dk0, d[2 * F + c0] = dv0;
    if (h1) d[F + c1] = dk1, d[2 * F + c1] = dv1;
}

}  // namespace

extern "C" int erc_rgcn_mean_fwd(const float* x, int ldx, int F, int R, int N, const int32_t* in_ptr,
                                 const int32_t* in_src, const int32_t* in_typ, float* Mout, int ldm, float* inv_cnt,
                                 void* stream) {
    ERC_REQUIRE(x && in_ptr && in_src && in_typ && Mout && inv_cnt, "rgcn_mean_fwd: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && R > 0 && R <= MAX_R && N > 0, "rgcn_mean_fwd: F=%d R=%d N=%d unsupported", F, R, N);
    ERC_REQUIRE(ldm >= (R + 1) * F && ldx >= F, "rgcn_mean_fwd: leading dimension too small");
    hipLaunchKernelGGL(rgcn_mean_fwd_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, F, R, N,
                       in_ptr, in_src, in_typ, Mout, ldm, inv_cnt);
    ERC_LAUNCH_CHECK("rgcn_mean_fwd");
    return ERC_OK;
}

extern "C" int erc_rgcn_mean_bwd(const float* dM, int ldm, int F, int R, int N, const int32_t* out_ptr,
                                 const int32_t* out_dst, const int32_t* out_typ, const float* inv_cnt, float* dx,
                                 int lddx, void* stream) {
    ERC_REQUIRE(dM && out_ptr && out_dst && out_typ && inv_cnt && dx, "rgcn_mean_bwd: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && R > 0 && R <= MAX_R && N > 0, "rgcn_mean_bwd: F=%d R=%d N=%d unsupported", F, R, N);
    hipLaunchKernelGGL(rgcn_mean_bwd_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, dM, ldm, F, R, N,
                       out_ptr, out_dst, out_typ, inv_cnt, dx, lddx);
    ERC_LAUNCH_CHECK("rgcn_mean_bwd");
    return ERC_OK;
}

extern "C" int erc_tconv_attn_fwd(const float* qkvs, int ld, int F, int N, float scale, const int32_t* in_ptr,
                                  const int32_t* in_src, float* out, int ldo, float* alpha, void* stream) {
    ERC_REQUIRE(qkvs && in_ptr && in_src && out && alpha, "tconv_attn_fwd: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && N > 0 && ld >= 4 * F, "tconv_attn_fwd: F=%d N=%d ld=%d unsupported", F, N, ld);
    hipLaunchKernelGGL(tconv_fwd_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, qkvs, ld, F, N, scale,
                       in_ptr, in_src, out, ldo, alpha);
    ERC_LAUNCH_CHECK("tconv_attn_fwd");
    return ERC_OK;
}

extern "C" int erc_tconv_attn_bwd_target(const float* qkvs, int ld, int F, int N, float scale, const int32_t* in_ptr,
                                         const int32_t* in_src, const float* alpha, const float* dout, int lddo,
                                         float* dqkvs, float* dscore, void* stream) {
    ERC_REQUIRE(qkvs && in_ptr && in_src && alpha && dout && dqkvs && dscore, "tconv_attn_bwd_target: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && N > 0 && ld >= 4 * F, "tconv_attn_bwd_target: bad sizes");
    hipLaunchKernelGGL(tconv_bwd_target_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, qkvs, ld, F, N,
                       scale, in_ptr, in_src, alpha, dout, lddo, dqkvs, dscore);
    ERC_LAUNCH_CHECK("tconv_attn_bwd_target");
    return ERC_OK;
}

extern "C" int erc_tconv_attn_bwd_source(const float* qkvs, int ld, int F, int N, const int32_t* out_ptr,
                                         const int32_t* out_dst, const int32_t* out_eid, const float* alpha,
                                         const float* dscore, const float* dout, int lddo, float* dqkvs,
                                         void* stream) {
    ERC_REQUIRE(qkvs && out_ptr && out_dst && out_eid && alpha && dscore && dout && dqkvs,
                "tconv_attn_bwd_source: null pointer");
    ERC_REQUIRE(F > 0 && F <= 128 && N > 0 && ld >= 4 * F, "tconv_attn_bwd_source: bad sizes");
    hipLaunchKernelGGL(tconv_bwd_source_kernel, dim3(erc_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, qkvs, ld, F, N,
                       out_ptr, out_dst, out_eid, alpha, dscore, dout, lddo, dqkvs);
    ERC_LAUNCH_CHECK("tconv_attn_bwd_source");
    return ERC_OK;
}
