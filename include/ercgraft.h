/*
 * libercgraft -- C-ABI of the MI355X-native ERC conversation-graph hot path.
 *
 * The reference (sailist/emotion-recognition-in-conversation) is pure Python:
 * it has no FFI / operator registry, its plugin boundary is the Python
 * convention "XModule.forward(**batch) -> (logits, aux)" (SURVEY.md 8b).  This
 * header is the boundary UNDERNEATH that convention: every entry point replaces
 * a chain of PyTorch / torch_geometric calls of the reference, cited per
 * function as file:line under /root/reference.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - the CALLER owns all memory (inputs, outputs, workspaces); no entry point
 *     allocates, frees or synchronises; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*), so calls are HIP-graph capturable.
 *   - return value: 0 on success, negative on error (ERC_E_*);
 *     erc_last_error() returns a static description of the last failure of the
 *     calling thread.  Nothing throws.
 *   - matrices are dense row-major fp32 with an explicit leading dimension
 *     unless stated otherwise; index arrays are int32; the drop-in edge lists
 *     (edge_index / edge_type) are int64 like the reference's.
 */
#ifndef ERCGRAFT_H
#define ERCGRAFT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ERC_ABI_VERSION 3

#define ERC_OK 0
#define ERC_E_ARG (-1)     /* bad shape / null pointer / unsupported size */
#define ERC_E_LAUNCH (-2)  /* hipLaunch failed */

int erc_abi_version(void);
const char* erc_last_error(void);

/* ------------------------------------------------------------------------
 * K1  window-graph builder.
 * Replaces batch_graphify + edge_perms: track_mm/cogmen_utils.py:109-172
 * (COGMEN, wp=wf=5) and track_mm/dgcn_models.py:51-118 (DialogueGCN,
 * wp=wf=10); relation id rule track_mm/cogmen.py:124-129 /
 * cogmen_utils.py:131-137:  type(j->k) = 2*(s_j*S + s_k) + (j<k ? 0 : 1).
 * Edge j->k exists iff max(0,j-wp) <= k <= min(L-1,j+wf), j,k in one dialogue.
 *
 *   lengths   int64 [B]                 text_length
 *   speakers  int64, element (b,t) at speakers[b*spk_sb + t*spk_st]
 *             (so both [B,T] and [T,B] layouts are accepted)
 * outputs (capacities: n_cap >= N = sum(lengths), e_cap >= E)
 *   node_off  int32 [B+1]   exclusive prefix sum of lengths
 *   node_row  int32 [n_cap] node -> row b*T+t of the padded [B,T,.] block
 *   node_spk  int32 [n_cap]
 *   in_ptr    int32 [n_cap+1], in_src / in_typ int32 [e_cap]:
 *             CSR by TARGET; edges in canonical (target, then source) order --
 *             edge e of this CSR is edge e of edge_index / edge_type below
 *   out_ptr   int32 [n_cap+1], out_dst / out_typ / out_eid int32 [e_cap]:
 *             CSR by SOURCE; out_eid = position of that edge in the in-CSR
 *   edge_index int64 [2, e_cap] (row 0 = source j, row 1 = target k) and
 *   edge_type  int64 [e_cap]: the reference's tensors; either may be NULL
 *   counts    int32 [2] = {N, E}
 * Entries beyond N / E are left untouched.
 */
int erc_window_graph_build(const int64_t* lengths, const int64_t* speakers, int64_t spk_sb, int64_t spk_st,
                           int B, int T, int wp, int wf, int n_speakers, int n_cap, int e_cap,
                           int32_t* node_off, int32_t* node_row, int32_t* node_spk,
                           int32_t* in_ptr, int32_t* in_src, int32_t* in_typ,
                           int32_t* out_ptr, int32_t* out_dst, int32_t* out_typ, int32_t* out_eid,
                           int64_t* edge_index, int64_t* edge_type, int32_t* counts, void* stream);

/* ------------------------------------------------------------------------
 * K2  dense fp32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32: exact
 * fp32 fma chains), C[M,N] = opA(A)[M,K] * opB(B)[K,N].
 * Replaces every nn.Linear / torch.matmul of the path and their autograd
 * (e.g. track_mm/cogmen.py:103-105,116-122; dagerc.py:94,98-106).
 *
 *   a_kmajor = 0: A(m,k) = A[row(m)*lda + k], row(m) = a_gather ? a_gather[m] : m
 *   a_kmajor = 1: A(m,k) = A[k*lda + m]
 *   b_kmajor = 0: B(k,n) = B[n*ldb + k]            ("NT": nn.Linear weight [out,in])
 *   b_kmajor = 1: B(k,n) = B[row(k)*ldb + n], row(k) = b_gather ? b_gather[k] : k
 *   a_gather / b_gather (int32) implement the fused gather of the valid rows of
 *   the padded [B,T,D] feature block (node_row of K1); NULL = identity.
 *   ones_col = 1: B gets a virtual extra column N of ones; its result
 *   (sum_k opA(A)[m,k], the bias gradient of an nn.Linear whose dY^T is A)
 *   goes to bias_out[m].  ones_col = 2: A gets a virtual extra row M of ones;
 *   its result (sum_k opB(B)[k,n], the bias gradient when dY is B, i.e. for
 *   [in,out]-stored weights) goes to bias_out[n].
 *   split_k = S > 1: the K range is cut in S pieces, piece z writes its partial
 *   product to C + z*c_slab (and bias_out + z*bias_slab); the caller reduces
 *   (erc_slab_reduce).  bias / act require S == 1.
 *   epilogue (S == 1): v = acc (+ bias[n]); act 0 none, 1 relu,
 *     2: v = aux[m*ldaux+n] > 0 ? v*act_scale : 0   (backward of relu+dropout)
 *     3: relu then inverted dropout with keep-probability 1-drop_p, mask from
 *        the counter RNG keyed by (rng_state[0] = per-step offset,
 *        rng_state[1] = seed, element m*N+n); act_scale = 1/(1-drop_p)
 *   accumulate != 0: C += v instead of C = v.
 */
int erc_gemm_f32(const float* A, int lda, int a_kmajor, const int32_t* a_gather,
                 const float* B, int ldb, int b_kmajor, const int32_t* b_gather,
                 float* C, int ldc, int M, int N, int K,
                 int split_k, int64_t c_slab, int ones_col, float* bias_out, int64_t bias_slab,
                 const float* bias, int act, const float* aux, int ldaux, float act_scale,
                 float drop_p, const uint64_t* rng_state, int accumulate, void* stream);

/* erc_gemm_f32 dispatches on the output width: N <= 512 runs the register-streaming kernel (16x32 tile per
 * wavefront, fragments loaded straight from global memory, K split over the wavefronts of a workgroup), wider
 * outputs the LDS-tiled kernel.  The streaming kernel is also exported under its own name (same contract). */
int erc_gemm_f32_stream(const float* A, int lda, int a_kmajor, const int32_t* a_gather,
                        const float* B, int ldb, int b_kmajor, const int32_t* b_gather,
                        float* C, int ldc, int M, int N, int K,
                        int split_k, int64_t c_slab, int ones_col, float* bias_out, int64_t bias_slab,
                        const float* bias, int act, const float* aux, int ldaux, float act_scale,
                        float drop_p, const uint64_t* rng_state, int accumulate, void* stream);
/* Batched weight gradients (autograd of every nn.Linear / matmul weight behind loss.backward(),
 * track_mm/cogmen.py:187-189, dagerc.py:228-231, mmgcn.py:150-152, dgcn.py:127-129): n_desc independent products
 * C_i[M_i,N_i] = A_i^T B_i[gather_i] in ONE launch.  A_i [K,M_i] fp32 K-major; B_i [*,N_i] K-major, fp32 or bf16 (the
 * feature block), optionally row-gathered; bias strip: ones = 1 -> bias_out[m] = sum_k A[k][m], ones = 2 ->
 * bias_out[n] = sum_k B[k][n].  64 x 64 tiles, K split over `splits` workgroups per tile; partial tiles are summed in
 * split order by the last-arriving workgroup (deterministic).  `table` is a device array of n_desc 104-byte records
 *   { const float* A; const void* B; float* C; float* bias_out; const int32_t* b_gather;
 *     int32 lda, ldb, ldc, M, N, K, ones, b_bf16, splits, tiles_n, item_base, n_items, tile_base, vec;
 *     float scale (C = scale A^T B); int32 a_bf16 (A is bf16 -- with an fp32 B only); }
 * with tiles_n = ceil(N/64), n_items = ceil(M/64)*tiles_n*splits, item_base / tile_base = running sums over the
 * previous records, ceil(ceil(K/4)/splits)*4 <= erc_wgrad_max_k_per_split(), no empty split, and vec bit0/1/2 set
 * when 16-byte (bf16: 8-byte) vector access to A / B / C is legal (M resp. N, the pitch and the base all multiples of
 * 4 elements).  item_base: HOST copy of the records' item_base fields (passed to the kernel by value, 32 records
 * per launch); n_items = sum of the records' n_items; slabs: n_items * erc_wgrad_slab_floats() floats of scratch;
 * counters: one int32 per output tile, zero before the first launch (every launch leaves them zero). */
int erc_wgrad_table(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                    int32_t* counters, void* stream);
int64_t erc_wgrad_slab_floats(void);
/* erc_wgrad_table with the THREE-TERM BF16 SPLIT available: a record whose mma_bf16 field is 2 (fp32 operands, 16-byte
 * accessible) multiplies x = h + m + l (three bf16 terms = 24 significant bits per operand value) on
 * v_mfma_f32_16x16x32_bf16, six cross products accumulated in fp32 -- fp32-class results at ~2x the rate of the exact fp32
 * matrix-core instruction, which is the ceiling of MMGCN's weight gradients (track_mm/mmgcn_models.py:373-394 under
 * autograd).  Other records behave as in erc_wgrad_table. */
int erc_wgrad_table_x3(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                       int32_t* counters, void* stream);
/* The same for the COGMEN bf16 compute mode (csrc/wgrad_bf16.hip): every record is C = A^T B with A [K, M <= 128] and
 * B [K, N] both K-major and BF16 in memory (B optionally through a row gather), on v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation; wave tile 128 x 64, so the wide operand is read once.  Record layout (112 bytes, little endian):
 *   u64 A, B, C, bias_a, bias_b, b_gather, k_dev; i32 lda, ldb, ldc, M, N, K, ct, cvec, splits, tiles_n, item_base, n_items,
 *   tile_base, kind
 * k_dev (or 0): device int32 holding the true K <= K (capacity mode, see erc_cogmen_bwd_tile): rows beyond it are masked;
 * ct != 0 stores C transposed (C[n * ldc + m]); bias_a [M] / bias_b [N] = fp32 column sums of the operand values over k (or
 * NULL); lda % 8 == 0 and ldb % 4 == 0 with finite pad columns up to 8 ceil(M / 8) resp. 4 ceil(N / 4); tiles_n =
 * ceil(N / 64); n_items = tiles_n * splits; K / splits <= erc_wgrad_bf16_max_k_per_split(); at most 16 records. */
int erc_wgrad_bf16(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs, int32_t* counters,
                   void* stream);
/* test hook: bound of erc_wgrad_bf16_adam's wait for a tile's splits (<= 0: default); 1 = the first unsatisfied wait times out */
int erc_wgrad_bf16_set_spin_limit(int limit);
/* The same for LARGE K (N = 33 k nodes at B = 512): a workgroup's four wavefronts take four neighbouring column tiles over the
 * same k-steps instead of splitting K, so an A row reaches the CU once per four tiles.  wg_base: HOST array, first workgroup
 * of every record (ceil(tiles_n / 4) * splits each); in the records item_base = first SLAB (tiles_n * splits slabs per record)
 * and n_items = the record's number of workgroups; n_wgs = total workgroups. */
int erc_wgrad_bf16_wide(const void* table, int n_desc, const int32_t* wg_base, int n_wgs, float* slabs, int32_t* counters,
                        void* stream);
int64_t erc_wgrad_bf16_slab_floats(void);
/* diagnostic: phase stamps (10 ns ticks) of work item `item` (of record 0) of the following erc_wgrad_bf16 launches,
 * 16 x uint64 device memory; NULL switches them off (tools/wgrad_stamps.py) */
int erc_wgrad_bf16_set_stamps(uint64_t* stamps, int item);
int erc_wgrad_bf16_max_k_per_split(void);
int erc_wgrad_max_k_per_split(void);

/* Forward input projection on a bf16 feature block: C[M,N] = act(X[gather(m), :K] W[N,K]^T + bias), X bf16,
 * W either fp32 (rounded to bf16 while loaded) or a bf16 shadow copy (w_is_bf16; see erc_adam_step), fp32
 * accumulate (v_mfma_f32_16x16x32_bf16); act 0 | 1 (relu).
 * With a bf16 shadow, N <= 128, K <= 1536 and M >= 1024 (env ERC_PERSIST_MIN_M) the persistent form runs: one
 * workgroup per CU keeps its share of W in registers and loops over 16-row groups, every feature row read once.
 * The HBM-dominant kernel of the COGMEN step (nn.Linear(D,100), track_mm/cogmen.py:103-105,147). */
int erc_gemm_bf16a_stream(const void* X, int ldx, const int32_t* gather, const void* W, int ldw, int w_is_bf16,
                          float* C, int ldc, int M, int N, int K, const float* bias, int act, void* stream);

/* Same contract with bf16 operands for the big streamed operand (the padded
 * feature block): A or B given as bf16 (uint16 storage), the other operand is
 * converted from fp32 while staging; v_mfma_f32_16x16x32_bf16, fp32 accumulate.
 *   x_is_a != 0: A is bf16 (forward input projection);  else B is bf16 (wgrad).
 */
int erc_gemm_bf16x(const void* A, int lda, int a_kmajor, const int32_t* a_gather,
                   const void* B, int ldb, int b_kmajor, const int32_t* b_gather, int x_is_a,
                   float* C, int ldc, int M, int N, int K,
                   int split_k, int64_t c_slab, int ones_col, float* bias_out, int64_t bias_slab,
                   void* stream);

/* C[M, N] = A[M, K] B[N, K]^T, fp32 operands and result, computed on the bf16 matrix cores from a three-term split of every
 * operand value (x = h + m + l, six cross products, fp32 accumulate: fp32-class, ~2^-23 relative per product); 128 x 128
 * tiles.  Replaces the exact-fp32 erc_gemm_f32 for the two 16 GFLOP products of MMGCN's GCNII chain
 * (track_mm/mmgcn_models.py:373-394).  split_k > 1: split s writes its partial product to C + s * c_slab (erc_slab_reduce
 * adds them).  K, lda, ldb multiples of 4; A, B 16-byte aligned. */
int erc_gemm_x3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int split_k,
                int64_t c_slab, void* stream);

/* erc_gemm_f32_grouped(form 1) on the same three-term split: block (b, m) of [n_dlg * n_mod] blocks of pitch x pitch floats,
 * Blk[L, L] = A_rows[L, K] B_rows[L, K]^T over the node rows m * n_nodes + [node_off[b], node_off[b + 1]); L <= max_rows <= 128
 * (one tile per block); split_k slabs of c_slab floats.  MMGCN: dAdj = sum_l dg_l z_l^T with K = 64 * 200. */
int erc_gemm_x3_grouped(const float* A, int lda, const float* B, int ldb, float* C, int pitch, const int32_t* node_off,
                        int n_dlg, int n_mod, int n_nodes, int max_rows, int K, int split_k, int64_t c_slab, void* stream);

/* out[(i / n_cols)*ld_out + i % n_cols] = act( sum_{s<S} slabs[s*slab_stride + i] + (bias ? bias[i % n_cols] : 0) ),
 * i < numel; ld_out = 0 means contiguous (ld_out = n_cols).  act: 0 none, 1 relu, 4 = add the sum to out instead of
 * overwriting it. */
int erc_slab_reduce(const float* slabs, int S, int64_t slab_stride, const float* bias, int n_cols, int act,
                    float* out, int ld_out, int64_t numel, void* stream);

/* Batched form for weight gradients: job j reduces S[j] slabs of numel[j]
 * floats at ws + src[j] (stride = stride[j]) into dst + dst_off[j].
 * `jobs` is a device int64 [n_jobs,5] = {src, stride, S, numel, dst_off}. */
int erc_slab_reduce_batched(const float* ws, float* dst, const int64_t* jobs, int n_jobs, int64_t max_numel,
                            void* stream);

/* ------------------------------------------------------------------------
 * K3  relation-segmented mean aggregation (torch_geometric RGCNConv,
 * aggr='mean', call site track_mm/cogmen.py:65,71):
 *   M[i, r*F:(r+1)*F] = mean_{j in N_r(i)} x_j   (0 when N_r(i) is empty)
 *   M[i, R*F:(R+1)*F] = x_i                       (root term)
 * so that RGCNConv(x) = M @ [W_0;..;W_{R-1};W_root] + bias is one GEMM.
 * inv_cnt [N,R] receives 1/|N_r(i)| (0 if empty) for the backward.
 * One wavefront per target node; edges of relation >= R are ignored (PyG).
 * F <= 128, R <= 8.
 */
int erc_rgcn_mean_fwd(const float* x, int ldx, int F, int R, int N,
                      const int32_t* in_ptr, const int32_t* in_src, const int32_t* in_typ,
                      float* Mout, int ldm, float* inv_cnt, void* stream);
/* dx[j] = sum_{e=(j->i)} dM[i, typ_e*F:..] * inv_cnt[i,typ_e] + dM[j, R*F:..]  (gather over out-edges) */
int erc_rgcn_mean_bwd(const float* dM, int ldm, int F, int R, int N,
                      const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_typ,
                      const float* inv_cnt, float* dx, int lddx, void* stream);

/* ------------------------------------------------------------------------
 * K4  per-target segmented softmax attention (torch_geometric
 * TransformerConv(heads=1, root_weight=True), call site
 * track_mm/cogmen.py:66,72).  qkvs [N, 4F] holds q | k | v | skip per node.
 *   alpha_e = softmax_{e in in(i)} ( q_i . k_src(e) * scale ),
 *   out_i   = sum_e alpha_e v_src(e) + skip_i
 * alpha [E] is kept for the backward (in-CSR order).
 */
int erc_tconv_attn_fwd(const float* qkvs, int ld, int F, int N, float scale,
                       const int32_t* in_ptr, const int32_t* in_src,
                       float* out, int ldo, float* alpha, void* stream);
/* backward, target side: dq_i, dskip_i and dscore_e (= dL/d(q.k), scale folded in).
 * Optional BatchNorm-backward prologue (bn_x != NULL; the BatchNorm1d that follows the layer, cogmen.py:67,72): `dout`
 * then holds dY = dL/d(BatchNorm output) and the kernel derives the layer's output gradient itself,
 *   dout_i = gamma * rstd * (dY_i - bn_bwd[c] - xhat_i * bn_bwd[F + c]),  xhat = (bn_x - saved[c]) * saved[F + c],
 * writing it to dout_store [N, lddo] (the source-side pass reads it there) -- one launch less than
 * erc_bn_bwd_apply + this. */
int erc_tconv_attn_bwd_target(const float* qkvs, int ld, int F, int N, float scale,
                              const int32_t* in_ptr, const int32_t* in_src, const float* alpha,
                              const float* dout, int lddo, float* dqkvs, float* dscore,
                              const float* bn_x, int bn_ldx, const float* bn_gamma, const float* bn_saved,
                              const float* bn_bwd, float* dout_store, void* stream);
/* backward, source side: dk_j = sum_{e out of j} dscore_e q_dst, dv_j = sum alpha_e dout_dst */
int erc_tconv_attn_bwd_source(const float* qkvs, int ld, int F, int N,
                              const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                              const float* alpha, const float* dscore, const float* dout, int lddo,
                              float* dqkvs, void* stream);

/* ------------------------------------------------------------------------
 * K5  BatchNorm1d (+ LeakyReLU) over the N nodes of this rank
 * (track_mm/cogmen.py:67-68,72; nn.BatchNorm1d semantics: biased variance for
 * the normalisation, unbiased for running_var, momentum 0.1, eps 1e-5).
 *   training != 0: batch statistics; saved[0:F] = mean, saved[F:2F] = rstd;
 *                  running_mean / running_var updated in place.
 *   training == 0: running statistics.
 *   y = leaky_relu(xhat*gamma + beta, slope)      (slope = 1 -> no activation)
 * ws: >= erc_bn_ws_floats(F) floats of scratch.
 */
int64_t erc_bn_ws_floats(int F);
int erc_bn_lrelu_fwd(const float* x, int ldx, int N, int F, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, float slope,
                     int training, float* saved, float* y, int ldy, float* ws, void* stream);
/* dx, dgamma, dbeta from dy (gradient wrt y) ; recomputes xhat and the activation sign from x */
int erc_bn_lrelu_bwd(const float* x, int ldx, int N, int F, const float* gamma, const float* beta,
                     const float* saved, float slope, const float* dy, int lddy,
                     float* dx, int lddx, float* dgamma, float* dbeta, float* ws, void* stream);

/* ------------------------------------------------------------------------
 * S3  cross entropy (F.cross_entropy: track_mm/cogmen.py:185, mmgcn.py:147,
 * masked dagerc.py:225-226, class-weighted dgcn.py:124).
 *   row_map (int32 [n_rows] or NULL): logits row of sample i is row_map[i]
 *     (DAG-ERC keeps padded [B*T,C] logits; the mask becomes a row map)
 *   weight (float [C] or NULL): loss = sum w_y nll / sum w_y
 *   out: stats[0] = loss, stats[1] = #(argmax == y), stats[2] = sum of weights; stats must hold >= 256
 *   floats, zero-initialised once by the caller (per-workgroup partials + an arrival counter live there)
 *   dlogits (may be NULL): gradient, same row addressing as logits; rows that
 *     no sample maps to are NOT written (caller zero-fills when row_map != NULL)
 */
int erc_cross_entropy(const float* logits, int ld, int C, int n_rows, const int32_t* row_map,
                      const int64_t* labels, const float* weight, float grad_scale,
                      float* dlogits, int lddl, float* stats, void* stream);

/* ------------------------------------------------------------------------
 * K9  COGMEN's Transformer encoder block rnn.0 (track_mm/cogmen.py:94-102; contrib/nn.py:283-305: post-norm, ReLU,
 * ffn 2048, batch_first, no padding mask).  Its output is DISCARDED by the reference (cogmen.py:146-147); these
 * entry points serve the faithful-cost mode and are inference-mode math.
 *   erc_enc_gemm_bf16:     C[M,N] = A[M,K] W[N,K]^T + bias (+ReLU), bf16 operands (K contiguous), fp32 accumulate on the
 *                          matrix cores; C as fp32 and / or bf16 (either may be NULL).  K, lda, ldw multiples of 4.
 *   erc_enc_attention:     qkv bf16 [n_seq*S, 3D] (q | k | v) -> softmax(q k^T / sqrt(D/heads)) v per (sequence, head)
 *                          over all S positions, out bf16 [n_seq*S, D].  D / heads <= 256.
 *   erc_enc_add_layernorm: y = LayerNorm(a + b) * gamma + beta per row of width D <= 2048, as fp32 and bf16.
 *   erc_enc_to_bf16:       fp32 -> bf16 (RNE) copy.
 */
int erc_enc_to_bf16(const float* x, int64_t n, void* y, void* stream);
int erc_enc_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, float* C_f32, void* C_bf16,
                      int ldc, int M, int N, int K, int relu, void* stream);
int erc_enc_attention(const void* qkv, int n_seq, int S, int D, int heads, void* out, void* stream);
int erc_enc_add_layernorm(const float* a, const float* b, int D, int n_rows, const float* gamma, const float* beta,
                          float eps, float* y_f32, void* y_bf16, void* stream);

/* K9, training half (the chained COGMEN variant `transformer_out(encoder(x))` with the key-padding
 * mask; layer math contrib/nn.py:283-305 in training mode and its backward; csrc/encoder_train.hip).
 *   erc_enc_gemm_bf16_ex:        erc_enc_gemm_bf16 with an epilogue: 0 none | 1 dropout(drop_p, scale) after the
 *                                optional ReLU | 2 multiply by (mask_src[row, col] != 0) * scale (backward of
 *                                ReLU + dropout, read off the stored bf16 forward output).
 *   erc_enc_attention_train:     attention with key-padding mask (keys >= lengths[seq] masked; lengths may be NULL) and
 *                                dropout on the probabilities; S <= 128, head dim <= 256; nothing saved.
 *   erc_enc_attention_bwd:       dqkv (bf16 [n_seq*S, 3D]) from qkv and dout (bf16 [n_seq*S, D]); recomputes the
 *                                probabilities and the dropout decisions.
 *   erc_enc_add_layernorm_train: y = LN(a + dropout(b)); saved_sum [n_rows, D] = a + dropout(b), saved_stats = mean
 *                                [n_rows] | rstd [n_rows].
 *   erc_enc_layernorm_bwd:       dy = dy_a[dy_a_map ? map[row] : row] (zero where map < 0) + dy_b (nullable);
 *                                ds fp32 = d(a + dropout(b)); db_bf16 = ds through the dropout mask; partial
 *                                [erc_enc_layernorm_bwd_blocks(n_rows)][2][D] = per-workgroup (dgamma | dbeta) sums,
 *                                to be finished with erc_enc_colsum over [blocks, 2D].
 *   erc_enc_transpose_bf16:      YT[c][r] = bf16(X[r][c]), r < R, zero for R <= r < ldyt (ldyt % 4 == 0); optional
 *                                plain bf16 copy.  Weight gradients are NT products over the row axis.
 *   erc_enc_colsum:              out[c] = sum_r X[r][c] (fp32 or bf16 X), fixed summation order; ws:
 *                                erc_enc_colsum_ws_floats(C) floats.
 *   erc_enc_inverse_rows:        inv[node_row[i]] = i, -1 elsewhere (inv: n_rows int32). */
int erc_enc_gemm_bf16_ex(const void* A, int lda, const void* W, int ldw, const float* bias, float* C_f32, void* C_bf16,
                         int ldc, int M, int N, int K, int relu, int epilogue, const void* mask_src, int ld_mask,
                         float scale, float drop_p, const uint64_t* rng_state, uint64_t rng_stream, void* stream);
int erc_enc_attention_train(const void* qkv, int n_seq, int S, int D, int heads, const int64_t* lengths, float drop_p,
                            const uint64_t* rng_state, uint64_t rng_stream, void* out, void* stream);
int erc_enc_attention_bwd(const void* qkv, const void* dout, int n_seq, int S, int D, int heads, const int64_t* lengths,
                          float drop_p, const uint64_t* rng_state, uint64_t rng_stream, void* dqkv, void* stream);
int erc_enc_add_layernorm_train(const float* a, const float* b, int D, int n_rows, const float* gamma, const float* beta,
                                float eps, float drop_p, const uint64_t* rng_state, uint64_t rng_stream, float* y_f32,
                                void* y_bf16, float* saved_sum, float* saved_stats, void* stream);
int erc_enc_layernorm_bwd_blocks(int n_rows);
int erc_enc_layernorm_bwd(const float* dy_a, const int32_t* dy_a_map, const float* dy_b, const float* saved_sum,
                          const float* saved_stats, const float* gamma, int D, int n_rows, float drop_p,
                          const uint64_t* rng_state, uint64_t rng_stream, float* ds, void* db_bf16, float* partial,
                          void* stream);
int erc_enc_transpose_bf16(const void* X, int x_is_f32, int ldx, int R, int C, void* YT, int ldyt, void* plain, int ldp,
                           void* stream);
int64_t erc_enc_colsum_ws_floats(int C);
int erc_enc_colsum(const void* X, int x_is_bf16, int ldx, int R, int C, float* out, float* ws, void* stream);
int erc_enc_inverse_rows(const int32_t* node_row, int N, int32_t* inv, int n_rows, void* stream);

/* Training-mode BatchNorm1d statistics (track_mm/cogmen.py:67): column mean / rstd of x [N,F] into saved[0,F) /
 * saved[F,2F), running_mean / running_var updated with `momentum` (unbiased variance), one launch.
 * ws: erc_bn_batch_stats_ws_floats(F) floats, 8-byte aligned, zero before the first call. */
int64_t erc_bn_batch_stats_ws_floats(int F);
int erc_bn_batch_stats(const float* x, int ldx, int N, int F, float* running_mean, float* running_var,
                       float momentum, float eps, float* saved, float* ws, void* stream);

/* COGMEN classifier head, forward + loss + backward in one launch (track_mm/cogmen.py:67-68,72-73 BatchNorm1d +
 * leaky_relu; :116-122 cls = Linear, ReLU, Dropout, Linear; :185 F.cross_entropy; :187-188 their backward):
 *   H3 = lrelu(bn(H2; saved, gamma, beta), slope);  Z = dropout(relu(H3 W0^T + b0), drop_p);  logits = Z W3^T + b3
 *   dlogits, dZ (through the relu / dropout mask), dY = dL/d(bn output) = (dZ W0) * lrelu'
 *   bn_bwd[0,F) = mean_rows(dY), bn_bwd[F,2F) = mean_rows(dY * xhat);  dbeta = sum dY, dgamma = sum dY * xhat
 *   stats[0] = weighted mean loss, [1] = #correct, [2] = sum of sample weights.
 * H2 [n_rows, ldh]; H3, Z, dZ, dY [n_rows, F]; logits, dlogits [n_rows, C]; W0 [F,F], W3 [C,F] row-major.
 * F <= 100, F % 4 == 0, C <= 8, 16-byte aligned H2 / gamma / beta / saved / W0 / H3.  Dropout uses the counter RNG
 * of erc_gemm_f32 act = 3 (element index row * F + col; rng_state = {offset, seed}), so the mask is the one
 * a separate Linear launch would draw.  ws: erc_head_fused_ws_floats(n_rows) floats, zero before the first call. */
int64_t erc_head_fused_ws_floats(int n_rows);
int erc_head_fused(const float* H2, int ldh, int n_rows, int F, int C, const float* gamma, const float* beta,
                   const float* saved, float slope, const float* W0, const float* b0, const float* W3,
                   const float* b3, const int64_t* labels, const float* weight, float drop_p,
                   const uint64_t* rng_state, float* H3, float* Z, float* logits, float* dlogits, float* dZ,
                   float* dY, float* bn_bwd, float* dgamma, float* dbeta, float* stats, float* ws, void* H3b, void* Zb,
                   void* dZb, void* dlb, int ldb16, const int32_t* n_dev, const int32_t* label_rows, int lddl, void* stream);
/* H3, Z, dZ, dlogits may be NULL when the bf16 copies (H3b, Zb, dZb, dlb) are given: the weight-gradient launch reads those, the
 * fp32 ones are then not written (1.2 of the 2.4 KB the launch writes per row).
 * lddl: row pitch of dlogits in floats (0 = C; the split compute modes pass 8: 16-byte rows for erc_wgrad_split, pad columns
 * are left untouched and must be finite).
 * label_rows (or NULL): the label of row i is labels[label_rows[i]] -- labels kept in a resident store (or a padded [B, T]
 * block) are read through the node -> row map instead of being compacted per batch.
 * H3b / Zb / dZb [n_rows, ldb16 >= F] and dlb [n_rows, 8]: optional bf16 copies (all four or none) of H3, Z, dZ and dlogits,
 * the operands of the classifier's weight gradients in the bf16 compute mode (erc_wgrad_bf16); pad columns untouched. */

/* erc_head_fused with BatchNorm's batch statistics finalised inside (training mode of nn.BatchNorm1d, cogmen.py:67):
 * bn_part [bn_tiles][2F] floats = per-tile column sums (sum x | sum x^2) from erc_cogmen_fwd_tile (bn_fused = 2);
 * every workgroup adds the tiles in order, workgroup 0 writes saved [2F] = mean | rstd (an OUTPUT here) and updates
 * running_mean / running_var (momentum, unbiased variance).  defer_reduce != 0: no last arriver -- bn_bwd, dgamma, dbeta and
 * stats are NOT written here; the consumer adds the workgroup records of ws itself (erc_cogmen_bwd_tile, head_part). */
int erc_head_fused_bn(const float* H2, int ldh, int n_rows, int F, int C, const float* gamma, const float* beta,
                      float* saved, float slope, const float* W0, const float* b0, const float* W3,
                      const float* b3, const int64_t* labels, const float* weight, float drop_p,
                      const uint64_t* rng_state, float* H3, float* Z, float* logits, float* dlogits, float* dZ,
                      float* dY, float* bn_bwd, float* dgamma, float* dbeta, float* stats, float* ws,
                      const float* bn_part, int bn_tiles, float* running_mean, float* running_var, float momentum,
                      float eps, int defer_reduce, void* H3b, void* Zb, void* dZb, void* dlb, int ldb16, const int32_t* n_dev,
                      const int32_t* label_rows, int lddl, void* stream);
/* floats per workgroup record of erc_head_fused's workspace: [0,112) column sums of dY, [112,224) of dY * xhat, [224] loss
 * part, [225] hits, [226] sum of the sample weights; ceil(n_rows / erc_head_fused_rows_per_workgroup(n_rows)) records */
int erc_head_fused_part_floats(void);
/* diagnostic: resident workgroups per CU of the throughput form of the head (more than 8 192 rows), by the runtime's occupancy query */
int erc_head_rows_occupancy(void);
/* rows per workgroup (= per partial record) of erc_head_fused{,_bn} at n_rows rows: 16 up to 8 192 rows, 32 beyond
 * (ERC_HEAD_ROWS=16 / 32 forces one); the consumer of the deferred records (erc_cogmen_bwd_tile) is told
 * ceil(n_rows / this) records */
int erc_head_fused_rows_per_workgroup(int n_rows);

/* diagnostic: 8 x uint64 phase stamps (10 ns ticks) of the middle workgroup of the following erc_head_fused[_bn] launches;
 * NULL switches them off (tools/cogmen_stamps.py) */
int erc_head_set_stamps(uint64_t* stamps);

/* Elementwise part of BatchNorm1d's backward: dx = gamma * rstd * (dY - bn_bwd[c] - xhat * bn_bwd[F + c]). */
int erc_bn_bwd_apply(const float* x, int ldx, int N, int F, const float* gamma, const float* saved,
                     const float* bn_bwd, const float* dY, int lddy, float* dx, int lddx, void* stream);

/* Fused classifier tail: logits = Z W^T + b (W [C,F], C <= 8, F <= 128), cross entropy as above, and
 * dZ = (Z > 0 ? mask_scale : 0) * (dlogits W)  -- the last Linear of the head, F.cross_entropy and their backward
 * through the preceding ReLU(+inverted dropout, mask_scale = 1/(1-p)) in one launch (track_mm/cogmen.py:116-122,185;
 * dgcn_models.py:163-170).  stats: >= erc_head_ce_stats_floats(n_rows) floats, zero-initialised once. */
int64_t erc_head_ce_stats_floats(int n_rows);
int erc_head_ce(const float* Z, int ldz, int F, int C, int n_rows, const float* W, const float* bias,
                const int64_t* labels, const float* weight, float mask_scale, float* logits, int ldl,
                float* dlogits, int lddl, float* dZ, int lddz, float* stats, void* stream);

/* ------------------------------------------------------------------------
 * S4  optimizer over the flat live-parameter buffer (torch.optim.Adam /
 * AdamW, track_mm/cogmen.py:50,187-189; dagerc.py:39,230-231).
 *   state: device int64 [ERC_ADAM_STATE_WORDS = 4 + 512] = {step count, RNG offset, RNG seed, unused, then one private
 *   copy of the step count per workgroup of the launch (all equal between calls)}; the call increments state[0], state[1]
 *   (so state+1 is a valid rng_state) and the private copies.  Whoever sets the step count by hand (checkpoint load)
 *   writes it to state[0] AND to state[4 .. 4+512).
 *   clip_norm > 0: grads are scaled by min(1, clip_norm/(gnorm+1e-6)) where
 *   gnorm[0] was produced by erc_grad_norm (clip_grad_norm_ semantics).
 *   decoupled != 0 -> AdamW (p *= 1 - lr*wd) else L2 (g += wd*p).
 *   grad_scale multiplies g first (1/world_size after a sum all-reduce).
 *   bf16_shadow (optional): the updated p[shadow_off, shadow_off+shadow_n) is also written as bf16 -- the weight
 *   operand of erc_gemm_bf16a_stream stays in sync without a conversion launch.
 *   skip_flag (optional): device int32; when non-zero at launch time the whole update is skipped (parameters, moments
 *   and the step counter stay as they are): the DAG-ERC recurrence kernels raise it when an exchange timed out, i.e.
 *   when this step's gradients are invalid -- the step fails on the device, no host synchronisation.
 */
#define ERC_ADAM_STATE_WORDS (4 + 512)
int erc_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled,
                  float grad_scale, float clip_norm, const float* gnorm, int64_t* state,
                  void* bf16_shadow, int64_t shadow_off, int64_t shadow_n, const int32_t* skip_flag, void* stream);
/* COGMEN graph part, bf16 compute mode, as row-tile kernels (csrc/cogmen_fused.hip): GNN.forward of
 * track_mm/cogmen.py:61-74 -- torch_geometric RGCNConv(100,100,8 relations, mean) -> TransformerConv(100,100,heads=1)
 * (cogmen.py:65-66,71-72) up to the input of BatchNorm1d (cogmen.py:67) -- in one launch, and its backward in one.
 * The window graph (cogmen.py:153-154: wp = wf = 5) lets a tile of 16 nodes work from a halo of +-5 (forward) / +-15
 * (backward) rows; wp, wf <= 5 is required.  Dense products on v_mfma_f32_16x16x32_bf16, fp32 accumulate; weights
 * are read from bf16 shadows (ErcShadowTab below, mode 1 = MFMA fragment order) of these logical [n][k] operands (zero padded):
 *   WcatT [112][928 = 29 K blocks]: WcatT[o][r*100 + c] = conv1.weight[r][c][o], r = 8 -> conv1.root[c][o]
 *   Wq    [400][128 =  4 K blocks]: rows = [lin_query; lin_key; lin_value; lin_skip].weight
 *   WqT   [112][416 = 13 K blocks]: WqT[c][n] = Wq[n][c]
 *   Wb    [112][960 = 30 K blocks]: Wb[c][r*104 + o] = conv1.weight[r][c][o] (r = 8: root)
 * Forward outputs: Mb bf16 [N, ldmb >= 900] = [mean_r H0 | H0] and H1b bf16 [N, ldh1b >= 100] (operands of the weight
 * gradients), inv_cnt [N,8], QKVS fp32 [N,400], H2 [N, ldh2], alpha [E] (softmax weights per in-edge).
 * node_spk / n_speakers: erc_window_graph_build's speaker array and the speaker count it was built with; with two speakers
 * (ids 0 / 1, the reference's GNN(n_speakers = 2)) the relation means are differences of per-speaker prefix sums over
 * the tile's rows, otherwise an edge-by-edge gather.
 * bn_fused = 1: also the training-mode BatchNorm statistics of H2 (saved = mean | rstd, running statistics updated)
 * by the last workgroup to arrive; bn_fused = 2: only the per-tile column sums, as floats [tiles][200] from bn_ws + 2
 * doubles on (= the bn_part operand of erc_head_fused_bn, which finalises them without a last arriver);
 * bn_ws = erc_cogmen_fwd_tile_ws_doubles(N) doubles, zero before the first call.
 * health / events (both or neither): erc_health_roll folded into this launch -- the first of a training step that precedes
 * a reader of the health word (erc_wgrad_bf16_adam, erc_adam_step*). */
int64_t erc_cogmen_fwd_tile_ws_doubles(int n_nodes);
int erc_cogmen_fwd_tile(const float* H0, int ldh0, int n_nodes, int wp, int wf, const int32_t* in_ptr,
                        const int32_t* in_src, const int32_t* in_typ, const void* WcatT, const float* b1,
                        const void* Wq, const float* bq, float scale, void* Mb, int ldmb, float* inv_cnt, void* H1b,
                        int ldh1b, float* QKVS, float* H2, int ldh2, float* alpha, int bn_fused,
                        float* running_mean, float* running_var, float momentum, float eps, float* saved,
                        double* bn_ws, const int32_t* node_spk, int n_speakers, const int32_t* n_dev, int32_t* health,
                        int32_t* events, void* stream);
/* Backward of the same: dY [N,100] = dL/d(BatchNorm output) (erc_head_fused), BatchNorm's elementwise backward
 * (gamma, saved, bn_bwd as erc_bn_bwd_apply), TransformerConv backward (target and source side), dH1 = dQKVS Wq,
 * the transposed relation means and dH0 = dP [W_r^T].  Outputs fp32: dQKVS [N,400], dH1 [N,100], dH0 [N, lddh0] --
 * the operands of the weight gradients (cogmen.py:187-188).
 * head_part != NULL (with erc_head_fused_bn's defer_reduce): bn_bwd is an OUTPUT -- every workgroup adds the head
 * kernel's head_parts workgroup records (head_part_floats = erc_head_fused_part_floats() each) itself, in the same order,
 * and workgroup 0 writes bn_bwd, dgamma (= sum dY * xhat), dbeta (= sum dY) and stats {mean loss, #correct, weight sum}. */
int erc_cogmen_bwd_tile(const float* dY, const float* H2, int ldh2, int n_nodes, int wp, int wf, const float* gamma,
                        const float* saved, const float* bn_bwd, const float* QKVS, const float* alpha,
                        const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr, const int32_t* out_dst,
                        const int32_t* out_typ, const int32_t* out_eid, const float* inv_cnt, const void* WqT,
                        const void* Wb, float scale, void* dQKVS, void* dH1, void* dH0, int lddh0,
                        const int32_t* node_spk, int n_speakers, const float* head_part, int head_parts,
                        int head_part_floats, float* dgamma, float* dbeta, float* stats, int grads_bf16, int lddh1,
                        const int32_t* n_dev, void* stream);
/* SPLIT COMPUTE MODES (terms = 2 | 3; "f32x2" / "f32x3"): the same two launches at fp32-class accuracy -- what the reference's
 * fp32 GNN.forward / backward computes (track_mm/cogmen.py:61-74,187-188) -- on the bf16 matrix cores.  The weights are read
 * from `terms` bf16 PLANES per operand (ErcShadowTab with terms planes, *_plane elements apart: the bf16 expansion w = t0 + t1
 * (+ t2)), the activation tiles are expanded into term planes as they are staged in LDS, and every product accumulates the
 * term products of weight >= 2^(-8 (terms - 1)) in fp32.  The operands handed to the weight-gradient launch
 * (erc_wgrad_split) are FP32: Mf [N, ldmf >= 900], H1f [N, ldh1f >= 100]; dQKVS [N, 400], dH1 [N, lddh1], dH0 [N, lddh0].
 * Two-speaker graphs only (n_speakers == 2, the reference's GNN(n_speakers = 2)): a node's row of M / dP then holds five
 * non-empty blocks, which is what lets the term planes fit in LDS.  Everything else as erc_cogmen_fwd_tile / _bwd_tile. */
int erc_cogmen_fwd_tile_x(int terms, const float* H0, int ldh0, int n_nodes, int wp, int wf, const int32_t* in_ptr,
                          const int32_t* in_src, const int32_t* in_typ, const void* WcatT, int64_t catT_plane,
                          const float* b1, const void* Wq, int64_t q_plane, const float* bq, float scale, float* Mf,
                          int ldmf, float* inv_cnt, float* H1f, int ldh1f, float* QKVS, float* H2, int ldh2,
                          float* alpha, int bn_fused, float* running_mean, float* running_var, float momentum,
                          float eps, float* saved, double* bn_ws, const int32_t* node_spk, int n_speakers,
                          const int32_t* n_dev, int32_t* health, int32_t* events, void* stream);
int erc_cogmen_bwd_tile_x(int terms, const float* dY, const float* H2, int ldh2, int n_nodes, int wp, int wf,
                          const float* gamma, const float* saved, const float* bn_bwd, const float* QKVS,
                          const float* alpha, const int32_t* in_ptr, const int32_t* in_src, const int32_t* out_ptr,
                          const int32_t* out_dst, const int32_t* out_typ, const int32_t* out_eid, const float* inv_cnt,
                          const void* WqT, int64_t qT_plane, const void* Wb, int64_t wb_plane, float scale, float* dQKVS,
                          float* dH1, float* dH0, int lddh0, const int32_t* node_spk, int n_speakers,
                          const float* head_part, int head_parts, int head_part_floats, float* dgamma, float* dbeta,
                          float* stats, int lddh1, const int32_t* n_dev, void* stream);
/* n_dev (here, in erc_cogmen_fwd_tile and in erc_head_fused{,_bn}; NULL = off): CAPACITY MODE.  n_nodes / n_rows is then the
 * capacity the grid and the buffers are sized for and the true count (<= capacity) is read from device memory -- counts[0]
 * of erc_cogmen_project_graph -- so that ONE captured HIP graph serves every batch whose node count fits the capacity
 * (rows >= the true count are masked exactly like the partial last tile; erc_wgrad_bf16 takes the same word as k_dev). */
/* grads_bf16 != 0: dQKVS [N, 400], dH1 [N, lddh1], dH0 [N, lddh0] are written as bf16 (pitches in elements, pad columns
 * untouched): they are only ever operands of the bf16 weight-gradient products (erc_wgrad_bf16), so the rounding moves from
 * that kernel's loads to these stores.  Otherwise fp32 (lddh1 >= 100). */

/* COGMEN bf16 mode, first launch of the step (csrc/cogmen_project.hip): the input projection
 *   H0[n, :n_out] = X[row(n), :K] W^T + bias   (track_mm/cogmen.py:103-105,147: rnn.1 on the valid utterances; X bf16
 *   [B*T, ldx] = the padded feature block of ERCCollate, W bf16 [n_out, ldw] = the optimizer-maintained shadow)
 * AND every output of erc_window_graph_build (same arrays, same values: track_mm/cogmen_utils.py:109-172) computed from
 * lengths [B] / speakers [B, T] by the projection's own workgroups -- no separate graph-build launch, no node -> row map
 * read in front of the feature rows.  counts[0] / counts[1] = number of nodes / edges; nothing else is written when they
 * exceed n_cap / e_cap.  Lengths are clamped to [0, T].  Supported sizes: erc_cogmen_project_graph_ok() != 0
 * (n_out <= 112, 32 <= K <= 1536, K, ldx, ldw multiples of 4, B <= 2048); X, W 8-byte aligned. */
int erc_cogmen_project_graph_ok(int K, int n_out, int B, int ldx, int ldw);
int erc_cogmen_project_graph(const void* X, int ldx, const void* W, int ldw, const float* bias, float* H0, int ldh0,
                             int n_out, int K, const int64_t* lengths, const int64_t* speakers, int64_t spk_sb,
                             int64_t spk_st, int B, int T, int wp, int wf, int n_speakers, int n_cap, int e_cap,
                             int32_t* node_off, int32_t* node_row, int32_t* node_spk, int32_t* in_ptr, int32_t* in_src,
                             int32_t* in_typ, int32_t* out_ptr, int32_t* out_dst, int32_t* out_typ, int32_t* out_eid,
                             int32_t* counts, const int32_t* desc, void* stream);
/* SPLIT COMPUTE MODES (terms = 2 | 3; "f32x2" / "f32x3"): the same launch with the fp32 feature block the reference feeds
 * rnn.1 (track_mm/cogmen.py:103-105,147) -- X fp32 [rows, ldx], 16-byte aligned rows -- and W = `terms` bf16 planes [n_out, ldw],
 * w_plane elements apart (ErcShadowTab mode 0, terms planes: the bf16 expansion of rnn.1.weight).  Every A fragment is
 * expanded into `terms` bf16 terms in registers; the term products of weight >= 2^(-8 (terms - 1)) are accumulated in fp32 on
 * v_mfma_f32_16x16x32_bf16: H0 to 2^-17 (two terms) / 2^-25 (three) of the exact product's operands (csrc/split_dev.h). */
int erc_cogmen_project_graph_x(int terms, const float* X, int ldx, const void* W, int64_t w_plane, int ldw, const float* bias,
                               float* H0, int ldh0, int n_out, int K, const int64_t* lengths, const int64_t* speakers,
                               int64_t spk_sb, int64_t spk_st, int B, int T, int wp, int wf, int n_speakers, int n_cap,
                               int e_cap, int32_t* node_off, int32_t* node_row, int32_t* node_spk, int32_t* in_ptr,
                               int32_t* in_src, int32_t* in_typ, int32_t* out_ptr, int32_t* out_dst, int32_t* out_typ,
                               int32_t* out_eid, int32_t* counts, const int32_t* desc, void* stream);
/* desc (or NULL): RESIDENT mode.  The batch is a list of B dialogue slots of a feature store that lives in HBM: desc[b] =
 * length of slot b (0 = empty), desc[B + b] = its first row in the store; X = the store's [U, ldx] bf16 feature rows,
 * speakers = its [U] speaker ids (element stride spk_st), lengths is unused (may be NULL), node_row receives STORE rows (the
 * weight-gradient gather and erc_head_fused's label_rows read the store through it).  No padded [B, T, D] block exists:
 * a training step needs 2 B int32 of new input. */

/* diagnostic: phase stamps (10 ns ticks) of the middle workgroup of the following erc_cogmen_{fwd,bwd}_tile launches,
 * 8 x uint64 device memory; NULL switches them off (tools/cogmen_stamps.py) */
int erc_cogmen_set_stamps(uint64_t* stamps);

/* bf16 shadow ranges maintained by the optimizer launch (bf16 compute mode: the weight operands of the bf16
 * matrix-core products -- rnn.1.weight of the input projection and the packed copies the fused COGMEN kernels read,
 * erc_cogmen_fwd_tile / erc_cogmen_bwd_tile -- stay in sync with the fp32 masters without an extra launch).
 * Element i of the flat parameter range [src_off, src_off + n_el), idx = i - src_off, is split into digits
 *   d0 = idx % n0, d1 = (idx / n0) % n1, d2 = idx / (n0 n1)
 * which give its coordinates in the LOGICAL B operand of its product, n = d0 sn0 + d1 sn1 + d2 sn2 (output column)
 * and k = d0 sk0 + d1 sk1 + d2 sk2 (reduction index); it is written as bf16 to
 *   mode 0: shadow_base[dst_off + n * ld + k]                                  (row-major [n][k]; identity copies)
 *   mode 1: shadow_base[dst_off + (((n / 16) * ld + k / 32) * 64 + ((k % 32) / 8) * 16 + n % 16) * 8 + k % 8]
 *           = the fragment order of v_mfma_f32_16x16x32_bf16's B operand, ld = number of 32-deep K blocks: a
 *           wavefront's fragment of (column tile, K block) is one contiguous 1 KB run.
 * The table is a HOST struct; elements no range maps to keep whatever the buffer held (zero padding: clear it once). */
typedef struct ErcShadowDesc {
    int64_t src_off, n_el, dst_off;
    int32_t n0, n1, sn0, sn1, sn2, sk0, sk1, sk2, ld, mode;
    /* ABI 3: terms = 1 .. 3 bf16 PLANES of the same layout, plane t at + t * plane_stride elements, holding term t of the
     * parameter's bf16 expansion p = t0 + t1 (+ t2), t0 = bf16(p), t1 = bf16(p - t0), t2 = bf16(p - t0 - t1) -- the weight
     * operands of the split compute modes (f32x2 / f32x3: fp32-class products on the bf16 matrix cores). */
    int64_t plane_stride;
    int32_t terms, pad_;
} ErcShadowDesc;
typedef struct ErcShadowTab {
    int32_t n, flags;          /* descriptors in use, <= 8; flags: set by the library */
    ErcShadowDesc d[8];
} ErcShadowTab;
/* erc_adam_step with a shadow table instead of the single identity range. */
int erc_adam_step_tab(float* p, const float* g, float* m, float* v, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled,
                      float grad_scale, float clip_norm, const float* gnorm, int64_t* state,
                      void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host, const int32_t* skip_flag,
                      void* stream);
/* Rebuild every shadow range from the fp32 parameters p[0, n) (after loading a state dict; or per forward when no
 * optimizer maintains the shadows).  shadow_numel (here and above): bf16 elements behind shadow_base; a descriptor
 * whose largest destination index lies outside is rejected (ERC_E_ARG). */
int erc_shadow_refresh(const float* p, int64_t n, void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host,
                       void* stream);

/* One-shot gradient exchange fused into the optimizer launch (data parallel, latency class: COGMEN's 1.1 MB of gradients;
 * SURVEY.md 8e; opt-in, ERC_DP_P2P=1 -- the RCCL all-reduce stays the default and the only choice for > 524 288 parameters or
 * clip-norm).  Every rank allocates a PUBLISH buffer (2 * n_pad floats: two parities) and a FLAG array (world * 512 int32)
 * with erc_p2p_alloc, exchanges the 64-byte IPC handles out of band and maps its peers' with erc_p2p_open.
 * erc_adam_step_p2p then replaces all-reduce + erc_adam_step_tab: per 1024-element chunk a workgroup publishes the local
 * gradient (write-through), posts (step << 1 | health bit) into flags[rank][chunk] of every rank, waits for the same chunk
 * of every rank in its local flag array (bounded polls; a timeout raises the health word) and applies the update with the
 * RANK-ORDERED sum (bit-identical on all ranks) times grad_scale.  A set health bit on any rank makes every rank skip the
 * step.  epoch: device int64 [512], zero-filled once; health: the local health word (also the step's skip flag). */
typedef struct ErcP2P {
    int32_t world, rank, spin_limit, pad;
    void* pub[8];        /* publish buffers of ranks 0 .. world-1 as mapped in THIS process (pub[rank] = the local one) */
    void* flags[8];      /* flag arrays likewise */
    void* epoch;
    void* health;
    int64_t n_pad;       /* floats per parity of a publish buffer, >= n, multiple of 4 */
} ErcP2P;
int erc_p2p_alloc(int64_t bytes, void** ptr, void* handle64);
int erc_p2p_open(const void* handle64, void** ptr);
int erc_p2p_close(void* ptr);
int erc_p2p_free(void* ptr);
int erc_adam_step_p2p(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int decoupled, float grad_scale, int64_t* state, void* shadow_base,
                      int64_t shadow_numel, const ErcShadowTab* tab_host, const ErcP2P* x, void* stream);

/* The same launch WITH THE OPTIMIZER INSIDE (single-rank steps; `optim.step()` of track_mm/cogmen.py:189 disappears as a
 * launch).  The last reduction of a tile becomes a reduce-scatter among its S splits: every split publishes its partial
 * tile, waits (bounded; `health` is raised on a timeout) until all S have, and finishes the quads x with x % S == its index:
 * sums them in split order, writes the gradient and applies torch.optim.Adam / AdamW to the same elements of p / m / v
 * (flat buffers with the layout of g; every record's C / bias_a / bias_b must point into g[0, n)), bf16 shadows of the table
 * included.  A record of kind 1 (splits = tiles_n = n_items = 1) names a range C[0, M) of g that an EARLIER launch completed
 * (BatchNorm's scale / shift).  counters: n_tiles + 512 int32, zero-filled once and then owned by this entry point (tile
 * counters count up monotonically; the tail holds per-workgroup launch sequence numbers: always launch the SAME table
 * with it).  `health` doubles as erc_adam_step_tab's skip_flag.  state / grad_scale / the shadow table as
 * erc_adam_step_tab; at most 256 work items (all of them resident); every record's split count a power of two <= 8, its C
 * 16-byte aligned with ldc % 4 == 0 and rows of whole quads (N % 4 == 0; M % 4 == 0 for ct records) -- a record that breaks
 * this raises `health`; no clip-norm. */
int erc_wgrad_bf16_adam(const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs, int32_t* counters,
                        int n_tiles, float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int decoupled, float grad_scale, int64_t* state, void* shadow_base,
                        int64_t shadow_numel, const ErcShadowTab* tab_host, int32_t* health, void* stream);

/* SPLIT COMPUTE MODES (terms = 2 | 3; "f32x2" / "f32x3"): erc_wgrad_bf16 / erc_wgrad_bf16_adam for FP32 operands -- the
 * autograd weight gradients of loss.backward() (track_mm/cogmen.py:187-188) at fp32-class accuracy on the bf16 matrix cores.
 * Same table records, but A [K, lda] and B [K or gathered rows, ldb] are fp32 in memory (lda, ldb multiples of 4, rows 16-byte
 * aligned, M and N multiples of 4); every value is expanded into `terms` bf16 terms in registers and the term products of
 * weight >= 2^(-8 (terms - 1)) are accumulated in fp32 (csrc/split_dev.h).  Slabs, counters, bias strips, capacity mode (k_dev)
 * and the fused optimizer exactly as the bf16 entry points (the wide form does not exist here). */
int erc_wgrad_split(int terms, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                    int32_t* counters, void* stream);
int erc_wgrad_split_adam(int terms, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                         int32_t* counters, int n_tiles, float* p, float* g, float* m, float* v, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                         int64_t* state, void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host,
                         int32_t* health, void* stream);

/* DATA PARALLEL with the gradient exchange INSIDE the weight-gradient + optimizer launch (ERC_DP_P2P=1; ErcP2P as
 * erc_adam_step_p2p; the reference gets its DDP all-reduce from accelerate: lumo/trainer/trainer.py:62-64,377-384).
 * erc_wgrad_bf16_adam (terms = 1) / erc_wgrad_split_adam (terms = 2 | 3) whose work items, having summed their quads over the
 * tile's splits, publish them at their flat gradient offsets, post (epoch, health bit) to every rank, wait (bounded) for the
 * same item of every rank and apply the update with the RANK-ORDERED sum times grad_scale (= 1 / world): bit-identical
 * replicas, no RCCL call, the N > 1 step stays 5 launches.  Every rank launches the SAME table (same work items); a health
 * bit on any rank vetoes the item's update on all.  x->health is the launch's health word; flag arrays: world * 512 int32;
 * x->epoch: int64 [512], zero-filled once.  UNVERIFIED ACROSS DEVICES (two processes on one GPU: tests/test_gpu_p2p.py). */
int erc_wgrad_adam_p2p(int terms, const void* table, int n_desc, const int32_t* item_base, int n_items, float* slabs,
                       int32_t* counters, int n_tiles, float* p, float* g, float* m, float* v, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                       int64_t* state, void* shadow_base, int64_t shadow_numel, const ErcShadowTab* tab_host,
                       const ErcP2P* x, void* stream);

/* Health word: one device int32 that the persistent kernels with bounded polls (erc_dag_rec_*, erc_gcnii_chain_*: their
 * `health` argument; NULL = use state[0] as before) raise to ERC_HEALTH_RAISED when a poll ran into its bound, i.e. when
 * this step's results are invalid.  The value is the bit pattern of 1.0f: the host side keeps the word in the tail of the
 * flat fp32 gradient buffer, so that under data parallelism it travels in the step's ONE gradient all-reduce and every
 * rank sees a non-zero word if any rank raised it.  erc_adam_step{,_tab}(skip_flag = the word) then skips the update on
 * every rank alike.  erc_health_roll, launched at the start of the next step, adds 1 to events[0] and clears the word if
 * it is still raised: a timeout costs that one step, the following steps run normally, and the host reads the event
 * count once per epoch. */
#define ERC_HEALTH_RAISED 0x3f800000
int erc_health_roll(int32_t* health, int32_t* events, void* stream);
/* diagnostic: out[0] = shader cycles, out[1] = 10-ns ticks of a fixed dependent-MFMA loop (bench.py --clock_probe) */
int erc_clock_probe(uint64_t* out, int iters, void* stream);
/* gnorm[0] = ||g * grad_scale||_2 ; ws >= 1024 floats */
int erc_grad_norm(const float* g, int64_t n, float grad_scale, float* gnorm, float* ws, void* stream);

/* ------------------------------------------------------------------------
 * K8  MMGCN's 64-layer GCNII chain (mmgcn_models.py:373-394, GraphConvolution.forward :27-39; nlayers 64, nhidden 200,
 * lamda 0.5, alpha 0.1, variant) as one persistent launch per direction (csrc/gcnii_chain.hip).
 *   out_l = theta_l [A h_l | h0] W_l + (1 - theta_l)((1 - alpha) A h_l + alpha h0) is re-associated to A (h_l V_l) + h0 U_l with
 *   V_l = theta_l W_l[:200] + (1 - theta_l)(1 - alpha) I, U_l = theta_l W_l[200:] + (1 - theta_l) alpha I, theta_l = ln(lamda / l + 1).
 *   erc_gcnii_chain_prep: W = convs.0.weight, layer l at W + l * w_stride ([400,200] each) -> VT [65][200][208] (row n,
 *     contiguous k; zero-filled once by the caller, the last plane is padding), V [65][200][208] (row k, contiguous n),
 *     U [200][64*200] (layer l at column l * 200), UT (or NULL) [64*200][200] = U transposed.  The caller computes
 *     Call = H0 U ([Mo*N][64*200]) with one GEMM (erc_gemm_x3 on UT, or erc_gemm_f32 on U).
 *   erc_gcnii_chain_config: parts per (dialogue, modality) block, rows per workgroup (<= 32) and dialogues per launch so that
 *     every workgroup of a dialogue is resident (occupancy query); T <= 128.
 *   erc_gcnii_chain_fwd: HD planes [66][Mo*N][200] (plane stride hd_plane): plane 1 = dropout(relu(fc0 x)) on entry, planes
 *     2..65 written (plane l+1 = dropout(relu(out_l)), dropout stream rng_stream0 + l as erc_gcnii_layer_fwd); ZS [Mo*N][lds]
 *     receives z_l = h_l V_l at column (l-1)*200.  ADJ / CR / node_off as erc_gemm_f32_grouped.
 *   erc_gcnii_chain_bwd: dHin = gradient wrt plane 65, dHout = gradient wrt plane 1; saves dg_l = d out_l -> DG and
 *     dz_l = A dg_l -> DZ (same layout as ZS).  Left to the caller (they only meet in sums over the layers):
 *     dW_l[:200] = theta_l HD_l^T dz_l, dW_l[200:] = theta_l H0^T dg_l, dH0 = DG U^T, dADJ = sum_l dg_l z_l^T (blocks, cross).
 *   ZX: exchange buffer [2][Mo*N][200]; state: int32 [1 + B + B*Mo*parts], zero-filled once ([0] error flag when health is
 *   NULL, epochs, flags); health: the health word (see erc_health_roll) a poll timeout raises. */
int erc_gcnii_chain_prep(const float* W, int64_t w_stride, float lamda, float alpha, float* VT, float* V, float* U, float* UT,
                         void* stream);
int erc_gcnii_chain_config(int B, int T, int Mo, int P, int* parts, int* rows, int* dialogues_per_launch);
int erc_gcnii_chain_fwd(const float* ADJ, int P, const float* CR, const int32_t* node_off, int N, int Mo, int B, int T,
                        int parts, int rows, int dialogues_per_launch, const float* VT, const float* Call, int ldc,
                        float* HD, int64_t hd_plane, float* ZS, int lds, float* ZX, int32_t* state, int32_t* health,
                        float drop_p, const uint64_t* rng_state, uint64_t rng_stream0, void* stream);
int erc_gcnii_chain_bwd(const float* ADJ, int P, const float* CR, const int32_t* node_off, int N, int Mo, int B, int T,
                        int parts, int rows, int dialogues_per_launch, const float* V, const float* HD, int64_t hd_plane,
                        const float* dHin, float* dHout, float* DG, float* DZ, int lds, float* ZX, int32_t* state,
                        int32_t* health, float drop_p, void* stream);
/* diagnostic: phase stamps (s_memtime) of workgroup 0 of the following launches, [64][16] uint64; NULL switches them off */
int erc_gcnii_chain_set_stamps(uint64_t* stamps);
/* test hook: bound of the chain kernels' exchange polls for the following launches (<= 0: the default, 4 000 000) */
int erc_gcnii_chain_set_spin_limit(int limit);

/* ------------------------------------------------------------------------
 * K6  DAG-ERC (track_mm/dagerc.py:109-189, track_mm/dagerc_models.py:312-365).
 *
 * erc_dag_meta: speaker ids (argmax of the one-hot speaker tensor, or integer
 * ids), DAG predecessor pred[b,t] = largest j < t with the same speaker (-1 if
 * none) -- row t of get_adj_v1's matrix (dagerc.py:109-129, windowp = 1) is
 * ones on [max(pred,0), t-1]; get_s_mask (dagerc.py:131-154) is spk[b,i]==spk[b,j].
 * Runs over the padded length T like the reference (pad = speaker 0).  Also
 * emits node_off [B+1] / node_row [N] (valid row b*T+t per utterance) used as
 * the row map of the masked cross entropy (dagerc.py:225).
 *   speaker_onehot: float, element (b,t,c) at [b*spk_sb + t*spk_st + c]; or
 *   speaker_ids: int64, element (b,t) at [b*spk_sb + t*spk_st].
 */
int erc_dag_meta(const float* speaker_onehot, const int64_t* speaker_ids, int64_t spk_sb, int64_t spk_st,
                 int n_speakers, const int64_t* lengths, int B, int T,
                 int32_t* spk, int32_t* pred, int32_t* node_off, int32_t* node_row, void* stream);

/* ------------------------------------------------------------------------
 * K6: the recurrence (dagerc.py:167-189, dagerc_models.py:326-365), hidden size 300, per layer and step t:
 *   M_t  = sum_{j in [max(pred,0), t-1]} softmax_j(w_q.H_l[t] + w_k.h_j + b) * (same speaker ? Wr0 h_j : Wr1 h_j)
 *   h_t  = GRUCell_c(x = H_l[t], h = M_t) + GRUCell_p(x = M_t, h = H_l[t])
 * weight-stationary with the
 * batch of dialogues as the MFMA M dimension (csrc/dag_rec.hip).  A group of `dg` dialogues is advanced by 300 / epc
 * workgroups; workgroup c keeps the weight rows / columns of its hidden elements [c epc, (c+1) epc) in registers for all
 * T steps and the groups' 300-vectors are exchanged as tagged 8-byte records (two exchanges per step and direction).
 *   erc_dag_rec_config(dir, ...): picks cfg[4] = {epc, dg, groups per launch, layers per launch} for direction dir
 *     (0 forward, 1 backward), B dialogues of padded length T and n_layers layers from the device's CU count and the
 *     occupancy query of the kernel, so that every workgroup of a launch is resident (hints > 0 force a value); when the
 *     device cannot hold everything at once the entry points below issue several launches.
 *   FORWARD: the layers of a launch run as a PIPELINE (layer l needs at its step t only h^{(l-1)}_{t+1}), each on its
 *     own workgroups: L x T dependent steps become T + 2 (L - 1).  The former hoisted GEMM of a layer -- GI = [W_ih(grus_c)
 *     H_l + b | W_hh(grus_p) H_l + b | w_q.H_l + b_lin] -- is computed by the layer's workgroups from the records the layer
 *     below publishes and SAVED to GI [B*T, ldgi >= 1801] (columns [0,1800) gate pre-activations, column 1800 the query
 *     score).  Per-layer operands are passed as host arrays of n_layers device pointers:
 *       Wh [1801(+),300] = W_ih(grus_c) ; W_hh(grus_p) ; w_q (row 1800), bh [1801] its biases (gather.linear.bias last),
 *       W_hh_c / b_hh_c = grus_c.weight_hh / bias_hh, W_ih_p / b_ih_p = grus_p.weight_ih / bias_ih, Wr = [Wr0 ; Wr1],
 *       w_k [300] = the key half of gather.linear.weight;  outputs H1 (row pitch ldo) and, saved for the backward, GI, Mseq [B*T,300], GH [B*T,1800] (sequential gate
 *       pre-activations), R [B*T,600] (Wr0 h | Wr1 h), ks [B*T] (w_k.h), alpha [B,T,T].  H0 [B*T, ldh0 >= 300] is the input of layer 0 (relu(fc1 x)).
 *   BACKWARD: the same pipeline, top layer first.  dHall [B*T, ldd >= 300 (n_layers + 1)] holds on entry the head's part
 *     of dL/d[H_0 | H_1 | .. | H_L] (block l+1 = the outputs of layer l) and receives in block 0 the complete gradient wrt
 *     H_0 through fc1's relu mask (H_0 > 0); the input gradient of a layer -- z_p g + [W_ih_c ; W_hh_p ; w_q]^T [dgates ;
 *     dqs], the former "dH_l += DGI Wh" GEMM -- is computed by the layer's workgroups and handed to the layer below per step.
 *     Written per layer: DGI [B*T, lddgi >= 1801] (columns [0,1800) hoisted-side gate gradients, column 1800 d(query
 *     score): d[W_ih_c ; W_hh_p ; w_q] = DGI^T H_l), DGH [B*T,1800] (d[W_hh_c ; W_ih_p] = DGH^T Mseq), dM [B*T,300]
 *     (d[Wr0 ; Wr1] = dM^T A with A from erc_dag_attn_sums), dks [B*T] (dw_k = dks^T H1).  Hl[l] = the input of layer l.
 *   state: int32 [1 + ceil(B / dg)], zero-filled ONCE by the caller: [0] is raised when a poll ran into its bound (results
 *     invalid; pass it to erc_adam_step as skip_flag) unless `health` names another word (erc_health_roll), then one
 *     launch epoch per group (shared by both directions).
 *   scratch: erc_dag_rec_scratch_bytes(dir, B, T, cfg) bytes per direction, 8-byte aligned, zero-filled ONCE.
 * T <= 1022 and the per-workgroup LDS (histories of the slice: grows with dg * T) <= 160 KB. */
int erc_dag_rec_config(int dir, int B, int T, int n_layers, int epc_hint, int dg_hint, int lpl_hint, int* cfg);
int64_t erc_dag_rec_scratch_bytes(int dir, int B, int T, const int* cfg);
/* diagnostic: while set, the recurrence kernels store shader-clock stamps of workgroup 0 per step and phase into
 * stamps[T][2][8] (tools/dag_stamps.py); NULL (the default) switches it off. */
int erc_dag_rec_set_stamps(uint64_t* stamps);
int erc_dag_rec_fwd(const float* H0, int ldh0, int n_layers, const float* const* Wh, const float* const* bh,
                    const float* const* W_hh_c, const float* const* b_hh_c, const float* const* W_ih_p,
                    const float* const* b_ih_p, const float* const* Wr, const float* const* w_k,
                    const int32_t* pred, const int32_t* spk, int B, int T, float* const* H1, int ldo,
                    float* const* GI, int ldgi, float* const* Mseq, float* const* GH, float* const* R,
                    float* const* ks, float* const* alpha, const int* cfg, int32_t* state, int32_t* health, void* scratch,
                    void* stream);
int erc_dag_rec_bwd(int n_layers, const float* const* Hl, int ldh, const float* const* GI, int ldgi,
                    const float* const* GH, const float* const* Mseq, const float* const* R, const float* const* alpha,
                    const float* const* Wh, const float* const* W_hh_c, const float* const* W_ih_p,
                    const float* const* Wr, const float* const* w_k, const int32_t* pred, const int32_t* spk, int B, int T,
                    float* dHall, int ldd, float* const* DGI, int lddgi, float* const* DGH, float* const* dM,
                    float* const* dks, const int* cfg, int32_t* state, int32_t* health, void* scratch, void* stream);
/* A[i, sel*300 + k] = sum_{j in window(i), relation sel} alpha[i,j] H1[j,k]  (sel 0: same speaker): the attention-weighted
 * sums of the hidden states, [B*T, 600].  d[Wr0 ; Wr1] = dM^T A (dagerc_models.py:356-358 under autograd). */
int erc_dag_attn_sums(const float* alpha, const float* H1, int ldo, const int32_t* pred, const int32_t* spk, int B, int T,
                      float* A, void* stream);



/* diagnostic: shader-clock stamps (5 x uint64) inside one step of workgroup (0,0) of the next forward scans; NULL = off */
int erc_lstm_set_stamps(unsigned long long* stamps);
/* ------------------------------------------------------------------------
 * (Bi)LSTM recurrence, hidden 100 per direction, torch.nn.LSTM semantics (gate order i|f|g|o):
 * DialogueGCN SeqContext (packed; track_mm/dgcn_models.py:10-33) and MMGCN's text branch (unpacked over
 * the padded length; track_mm/mmgcn.py:69,113-114).  One layer, both directions per call.
 *   GX [rows, >=800]: hoisted x W_ih^T + b_ih, direction d in columns [400d, 400d+400)  (one GEMM by the caller)
 *   W_hh [2][400,100], b_hh [2][400]: weight_hh_l{k}, weight_hh_l{k}_reverse and their biases
 *   lengths: int64 [B] (packed: dialogue b runs L_b steps, the reverse direction starts at L_b-1; positions
 *            >= L_b produce zeros) or NULL (every dialogue runs T steps)
 *   rows: row(b,t) = b*sb + t*st (in rows), or node_off[b] + t when node_off != NULL
 *   Hout [rows, ldh] columns [100d,100d+100); Hdrop (optional): the same with inverted dropout(drop_p)
 *   applied -- the inter-layer dropout of nn.LSTM; mask keyed by (rng_state, rng_stream, element)
 *   saved for the backward: gates [rows,800] (post-activation), Cst [rows,200], Hprev [rows,200] (h_{t-1} in
 *   scan order: dW_hh[d] = dGX[:,400d:]^T Hprev[:,100d:])
 * Backward: dHout = gradient wrt Hout (wrt Hdrop when drop_p > 0); writes dGX [rows,800] (zero on padded
 * rows); dW_ih = dGX^T x, db = colsum(dGX), dx = dGX W_ih are GEMMs by the caller.
 */
int erc_lstm_scan_fwd(const float* GX, int ldgx, const float* W_hh, const float* b_hh, const int64_t* lengths,
                      const int32_t* node_off, int64_t sb, int64_t st, int B, int T,
                      float* Hout, int ldh, float* Hdrop, int ldhd, float drop_p, const uint64_t* rng_state,
                      uint64_t rng_stream, float* gates, float* Cst, float* Hprev, void* stream);
int erc_lstm_scan_bwd(const float* W_hh, const int64_t* lengths, const int32_t* node_off, int64_t sb, int64_t st,
                      int B, int T, const float* gates, const float* Cst, const float* dHout, int lddh,
                      float drop_p, const uint64_t* rng_state, uint64_t rng_stream, float* dGX, void* stream);

/* ------------------------------------------------------------------------
 * DialogueGCN graph operators (track_mm/dgcn_models.py:36-152, models/rgcn.py:264-355) over the CSRs of K1.
 */
/* dst[i,:] = src[map[i],:] (scatter = 0) or dst[map[i],:] = src[i,:] (scatter = 1): compaction of the valid rows of
 * a padded [B,T,F] block (node_features.append(features[j,:cur_len]), dgcn_models.py:67) and its transpose. */
int erc_gather_rows(const float* src, int lds, const int32_t* map, int N, int F, float* dst, int ldd, int scatter,
                    void* stream);
/* EdgeAtt (dgcn_models.py:121-152): att = x W^T is a GEMM by the caller; per SOURCE node j
 * norm[e] = softmax_{e out of j} ( att[dst_e] . x_j ), written in in-CSR (edge_index) order via out_eid. */
int erc_edge_att_fwd(const float* x, int ldx, const float* att, int lda, int F, int N,
                     const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid, float* norm, void* stream);
/* backward: dx (+)= sum_e dscore_e att[dst_e] per source, datt[k] = sum_{e into k} dscore_e x[src_e] per target;
 * dscore [E] is scratch (in-CSR order). */
int erc_edge_att_bwd(const float* x, int ldx, const float* att, int lda, int F, int N,
                     const int32_t* in_ptr, const int32_t* in_src,
                     const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                     const float* norm, const float* dnorm, float* dx, int lddx, int accumulate_dx,
                     float* datt, int ldda, float* dscore, void* stream);
/* the same with d norm given as dn_parts partial vectors dnorm[s * dn_stride + e] (the basis groups of
 * erc_brgcn_bwd_edges_tile), summed in order while they are read */
int erc_edge_att_bwd_parts(const float* x, int ldx, const float* att, int lda, int F, int N,
                           const int32_t* in_ptr, const int32_t* in_src,
                           const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                           const float* norm, const float* dnorm, int dn_parts, int64_t dn_stride, float* dx, int lddx,
                           int accumulate_dx, float* datt, int ldda, float* dscore, void* stream);
/* erc_edge_att_bwd_parts with two more jobs of DialogueGCN's backward inside its source-side launch (both optional):
 *  - dx_slabs != NULL: dx[j, c] += sum_s dx_slabs[s * dx_slab_stride + j * F + c], the partial feature gradients of
 *    erc_brgcn_bwd_source_tile (replaces erc_slab_reduce(mode 4));
 *  - rs_TT != NULL: rs_R extra workgroups form rs_datt[r, :30] = sum_{e: rs_typ[e] == r} rs_TT[e, :30], the relation sums
 *    behind erc_brgcn_bwd_edges_tile called with datt = NULL (d att, models/rgcn.py:330; rs_counts[1] = #edges). */
int erc_edge_att_bwd_fused(const float* x, int ldx, const float* att, int lda, int F, int N,
                           const int32_t* in_ptr, const int32_t* in_src,
                           const int32_t* out_ptr, const int32_t* out_dst, const int32_t* out_eid,
                           const float* norm, const float* dnorm, int dn_parts, int64_t dn_stride, float* dx, int lddx,
                           int accumulate_dx, float* datt, int ldda, float* dscore, const float* dx_slabs, int n_dx_slabs,
                           int64_t dx_slab_stride, const float* rs_TT, const int32_t* rs_typ, const int32_t* rs_counts,
                           float* rs_datt, int rs_R, void* stream);
/* basis-decomposed RGCNConv with edge_norm, add aggregation (models/rgcn.py:329-355), num_bases = 30:
 *   Z[i, b*F + c] = sum_{e into i} norm_e att[type_e, b] x[src_e, c]      so that
 *   conv(x) = Z @ basis.view(30F, out) + x @ root + bias                  (two GEMMs by the caller). */
int erc_brgcn_agg_fwd(const float* x, int ldx, int F, int N, const int32_t* in_ptr, const int32_t* in_src,
                      const int32_t* in_typ, const float* norm, const float* att, int num_bases, float* Z, void* stream);
/* edge-side backward from dZ = dOut @ basis.view(30F,out)^T: dnorm[e] (gradient into EdgeAtt), TT [E,30] scratch,
 * datt [R,30] = gradient of conv1.att (one workgroup per relation, fixed order); counts = {N,E} of K1. */
int erc_brgcn_bwd_edges(const float* x, int ldx, int F, int N, int R, const int32_t* in_ptr, const int32_t* in_src,
                        const int32_t* in_typ, const int32_t* counts, const float* norm, const float* att,
                        int num_bases, const float* dZ, float* dnorm, float* TT, float* datt, void* stream);
/* node-side backward: U[j, b*O + c] = sum_{e out of j} norm_e att[type_e,b] dOut[dst_e, c]; then
 * dx = U @ basisT with basisT [30*O, F] = erc_transpose_batched(basis) (one GEMM by the caller). */
int erc_brgcn_bwd_source(const float* dH, int lddh, int O, int N, const int32_t* out_ptr, const int32_t* out_dst,
                         const int32_t* out_typ, const int32_t* out_eid, const float* norm, const float* att,
                         int num_bases, float* U, void* stream);
/* The forward of the layer as ONE tile launch (F = 200, O = 100, 30 bases: dgcn_models.py:41-45): 16 nodes x 5 bases per
 * workgroup, Z blocks aggregated into LDS (and written to Z [N, 30F] for the weight gradient), multiplied by the basis
 * rows on the fp32 matrix cores; the root term rides along as a 6th block of the last basis group.  Leaves
 * S = erc_brgcn_fwd_tile_slabs() partial [N, O] slabs: conv(x) = erc_slab_reduce(slabs, S, N*O, bias). */
int64_t erc_brgcn_fwd_tile_slab_floats(int n_nodes);
int erc_brgcn_fwd_tile_slabs(void);
int erc_brgcn_set_stamps(unsigned long long* stamps);
/* The node side of the layer's backward as ONE tile launch (replaces erc_brgcn_bwd_source + erc_transpose_batched + the
 * [N, 30 O] x [30 O, F] GEMM + the root GEMM): per 16 source nodes x 5 bases, U blocks (sum over out-edges of
 * norm_e att[type_e, b] dOut[dst_e, :]) into LDS, times basis[b]^T on the fp32 matrix cores (weight fragments are 16-byte
 * loads ALONG k straight from basis [30, F, O] -- no transposed copy), dOut @ root^T as a 6th block of the last group.
 * Leaves S = erc_brgcn_fwd_tile_slabs() partial [N, F] slabs: dx += erc_slab_reduce(slabs, S, N*F, act = 4). */
/* The edge side of the backward without dZ in HBM (replaces the [N, O] x [O, 30 F] GEMM + erc_brgcn_bwd_edges): dZ blocks
 * on the matrix cores into LDS, 6 dot products per in-edge.  TT [E, 30] scratch, datt [R, 30] as erc_brgcn_bwd_edges;
 * d norm arrives as S = erc_brgcn_fwd_tile_slabs() partial vectors dn_slabs[s * dn_stride + e]:
 * dnorm = erc_slab_reduce(dn_slabs, S, dn_stride, numel = E). */
int erc_brgcn_bwd_edges_tile(const float* x, int ldx, int F, int O, int N, int R, const int32_t* in_ptr,
                             const int32_t* in_src, const int32_t* in_typ, const int32_t* counts, const float* norm,
                             const float* att, int num_bases, const float* basis, const float* dH, int lddh, float* TT,
                             float* dn_slabs, int64_t dn_stride, float* datt, void* stream);
int erc_brgcn_bwd_source_tile(const float* dH, int lddh, int F, int O, int N, const int32_t* out_ptr, const int32_t* out_dst,
                              const int32_t* out_typ, const int32_t* out_eid, const float* norm, const float* att,
                              int num_bases, const float* basis, const float* root, float* slabs, void* stream);   /* diagnostic: 6 x uint64 phase stamps (10 ns ticks); NULL = off */
int erc_brgcn_fwd_tile(const float* x, int ldx, int F, int O, int N, const int32_t* in_ptr, const int32_t* in_src,
                       const int32_t* in_typ, const float* norm, const float* att, int num_bases, const float* basis,
                       const float* root, float* Z, float* slabs, void* stream);
/* The same layer in RELATION space, for R <= erc_rrgcn_max_relations() (= 8: two speakers).  models/rgcn.py:300-304 composes
 * W_r = sum_b comp[r,b] basis[b] and transforms per relation; with R < num_bases that order is also the cheaper one:
 *   erc_basis_compose:   Wr [R,F,O] and its per-relation transpose WrT [R,O,F]
 *   erc_rrgcn_agg_fwd:   Z[i, r*F + c] = sum_{e into i, type_e = r} norm_e x[src_e, c];  conv(x) = Z @ Wr.view(RF,O) + ...
 *   erc_rrgcn_bwd_edges: dnorm[e] = x[src_e] . dZ[i, type_e, :]   from dZ = dOut @ Wr.view(RF,O)^T
 *   erc_rrgcn_bwd_source: U[j, r*O + c] = sum_{e out of j, type r} norm_e dOut[dst_e, c];  dx = U @ WrT.view(RO,F)
 *   erc_basis_decompose: dbasis[b] = sum_r comp[r,b] dWr[r], dcomp[r,b] = <dWr[r], basis[b]> from dWr = Z^T dOut [R,F*O]
 * (fixed summation orders).  Larger R (MELD: 162) keeps the basis-space entry points above. */
int erc_rrgcn_max_relations(void);
int erc_basis_compose(const float* comp, const float* basis, int R, int num_bases, int F, int O, float* Wr, float* WrT,
                      void* stream);
int erc_basis_decompose(const float* comp, const float* basis, const float* dWr, int R, int num_bases, int FO,
                        float* dbasis, float* dcomp, void* stream);
int erc_rrgcn_agg_fwd(const float* x, int ldx, int F, int N, int R, const int32_t* in_ptr, const int32_t* in_src,
                      const int32_t* in_typ, const float* norm, float* Z, void* stream);
int erc_rrgcn_bwd_edges(const float* x, int ldx, int F, int N, int R, const int32_t* in_ptr, const int32_t* in_src,
                        const int32_t* in_typ, const float* dZ, float* dnorm, void* stream);
int erc_rrgcn_bwd_source(const float* dH, int lddh, int O, int N, int R, const int32_t* out_ptr, const int32_t* out_dst,
                         const int32_t* out_typ, const int32_t* out_eid, const float* norm, float* U, void* stream);
/* out[n][c][r] = in[n][r][c] */
int erc_transpose_batched(const float* in, int nb, int rows, int cols, float* out, void* stream);
/* out[i,:] (+)= sum_{e in CSR row i} x[idx[e],:]: the neighbour sum of torch_geometric GraphConv(aggr='add')
 * (call site dgcn_models.py:42,46) with the in-CSR, its transpose with the out-CSR.  F <= 256. */
int erc_csr_sum(const float* x, int ldx, int F, int N, const int32_t* ptr, const int32_t* idx, float* out, int ldo,
                int accumulate, void* stream);

/* DialogueGCN's tail in ONE launch (csrc/dgcn_tail.hip): Hc = sum of the RGCN partial outputs slabs[s * slab_stride + i] + bias
 * (models/rgcn.py:345-355), GraphConv graph_out = W_rel sum_{j->i} Hc_j + b + W_root Hc_i written to Xc[:, 200:300)
 * (dgcn_models.py:42,46), Classifier lin1 + ReLU + inverted dropout + lin2 on [features | graph_out] (dgcn_models.py:163-170),
 * (class-weighted) F.cross_entropy (dgcn.py:124; stats as erc_head_ce) and the backward down to dXc [N,300] and the row-local
 * GraphConv gradients dAGG = dG W_rel, dHc = dG W_root (the scatter over the out-edges is erc_csr_sum(accumulate) afterwards).
 * Replaces erc_slab_reduce + erc_csr_sum + 3 forward GEMMs + erc_head_ce + 3 backward GEMMs.  The in-CSR's sources of a row lie
 * within `window` <= erc_dgcn_tail_max_window() rows of it; n_rows <= erc_dgcn_tail_max_rows(); n_classes <= 8; dropout draws
 * erc_uniform(seed, offset, row * 100 + column) like erc_gemm_f32_stream's act 3.  stats: >= erc_dgcn_tail_stats_floats(n_rows)
 * floats, zero-initialised once.  Exact fp32 products (v_mfma_f32_16x16x4_f32). */
int erc_dgcn_tail_max_rows(void);
int erc_dgcn_tail_set_stamps(uint64_t* stamps);   /* diagnostic: 16 x uint64 phase stamps (10 ns ticks) of workgroup 0; NULL = off */
int erc_dgcn_tail_max_window(void);
int64_t erc_dgcn_tail_stats_floats(int n_rows);
int erc_dgcn_tail(const float* slabs, int n_slabs, int64_t slab_stride, const float* rgcn_bias, const int32_t* in_ptr,
                  const int32_t* in_src, int window, const float* W_rel, const float* b_rel, const float* W_root,
                  const float* W1, const float* b1, const float* W2, const float* b2, const int64_t* labels,
                  const float* weight, int n_classes, int n_rows, float drop_p, const uint64_t* rng, float* Xc, int ldx,
                  float* Hc, float* AGG, float* Zc, float* logits, float* dlogits, float* dZc, float* dXc, int lddx,
                  float* dAGG, float* dHc, float* stats, void* stream);

/* ------------------------------------------------------------------------
 * MMGCN (track_mm/mmgcn.py:56-123, track_mm/mmgcn_models.py:8-39,344-394,493-646).  Node rows are
 * modality-major: node (m, i) = m*N + i, i = node_off[b] + t, in the modality order [a, v, l] of
 * mmgcn_models.py:586-593.  The (modalities*N)^2 adjacency of create_big_adj is kept as its non-zero
 * structure: blocks [B*M][P][P] (P >= max length, multiple of 4) + cross entries [B][M*M][P].
 */
/* Grouped block products on the matrix cores.  form 0: C_nodes[L,N] (+)= Blk[L,L] B_nodes[L,N] (A*h of
 * GraphConvolution.forward, mmgcn_models.py:29, and its transpose: the normalised adjacency is symmetric);
 * form 1: Blk[L,L] (+)= A_nodes[L,K] B_nodes[L,K]^T (cosine blocks of mmgcn_models.py:604-608, and dAdj).
 * cross (form 0, optional): the cross-modal entries [B][M*M][P] of the adjacency; their contribution
 * C[(m,t),:] += sum_{n != m} cross[b][m*M+n][t] * B_nodes[(n,t),:] is added in the epilogue (what
 * erc_mm_cross_apply does as a launch of its own). *
 * planes > 1 (form 1 and erc_mm_cross_grad, erc_gemm_f32_planes): the contraction also runs over `planes` operand planes
 * (plane p of A at A + p * a_plane elements, of B at B + p * b_plane): sum_p A_p B_p^T in one launch.  MMGCN's backward
 * uses it for the 64 per-layer contributions to the adjacency gradient (mmgcn_models.py:373-394 under autograd) and to
 * the gradient of h0, which only meet in a sum and therefore need not sit on the per-layer dependency chain.
 * split > 1 (form 1): the contraction range is cut into `split` parts, part s stores into slab C + s * c_slab
 * (reduced in slab order by erc_slab_reduce). */
int erc_gemm_f32_grouped(int form, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         int N_or_K, const int32_t* node_off, int n_dialogues, int n_mod, int n_nodes,
                         int max_len, int pitch, int accumulate, int act, const float* aux, int ldaux,
                         float act_scale,
                         const float* cross, int planes, int64_t a_plane, int64_t b_plane, int split, int64_t c_slab,
                         void* stream);
/* C[M,N] (+)= sum_p A_p[M,K] B_p[N,K]^T over `planes` planes (row-major operands, K contiguous); split_k > 1 writes
 * partial slabs of c_slab floats each (reduced by erc_slab_reduce) instead of C. */
int erc_gemm_f32_planes(const float* A, int lda, int64_t a_plane, const float* B, int ldb, int64_t b_plane, float* C,
                        int ldc, int M, int N, int K, int planes, int split_k, int64_t c_slab, int accumulate,
                        void* stream);
/* node tables from text_length and the time-major one-hot qmask [T,B,S] (element (t,b,c) at t*q_st + b*q_sb + c) */
int erc_mm_meta(const int64_t* lengths, const float* qmask, int64_t q_st, int64_t q_sb, int n_speakers, int B,
                int32_t* node_off, int32_t* node_row, int32_t* node_dlg, int32_t* node_spk, void* stream);
/* simple_batch_graphify (mmgcn_utils.py:5-21) [+ speaker embedding add, mmgcn_models.py:540-545]; F = 200 */
int erc_mm_flatten(const float* src, int lds, const int32_t* row_map, const float* emb, const int32_t* spk, int N,
                   float* dst, int ldd, void* stream);
int64_t erc_mm_emb_grad_ws_floats(int n_speakers);   /* floats of `ws` below */
int erc_mm_emb_grad(const float* dl, int ld, const int32_t* spk, int N, int n_speakers, float* demb, float* ws,
                    void* stream);
int erc_mm_row_normalize(const float* x, int R, float* xhat, float* inv, void* stream);
int erc_mm_row_normalize_bwd(const float* xhat, const float* inv, const float* dxhat, int R, float* dx, void* stream);
/* COS blocks (raw cosines, form-1 grouped GEMM of xhat) -> sim = 1 - acos(0.99999 cos)/pi, cross-modal
 * same-utterance sims, degrees over the full row, D^-1/2 A D^-1/2 (mmgcn_models.py:604-644). */
int erc_mm_adj_finish(const float* COS, const float* xhat, const int32_t* node_off, int B, int M, int N, int P,
                      float* ADJ, float* CR, float* CCOS, float* DEG, void* stream);
/* its backward: G = dCOS + dCOS^T per block (dXhat_block = G Xhat by a form-0 grouped GEMM), GC per cross pair;
 * DD [M*N] is scratch (the per-node degree gradients, written by the first of the two launches) */
int erc_mm_adj_finish_bwd(const float* COS, const float* CCOS, const float* DEG, const float* dADJ, const float* dCR,
                          const int32_t* node_off, int B, int M, int N, int P, float* G, float* GC, float* DD,
                          void* stream);
/* out[(m,i),:] += sum_{n != m} CR[b][m*M+n][p] h[(n,i),:] ;  dCR[b][m*M+n][p] += dhi[(m,i),:] . h[(n,i),:] */
int erc_mm_cross_apply(const float* CR, const float* h, int ldh, const int32_t* node_dlg, const int32_t* node_off,
                       int M, int N, int P, float* out, int ldo, void* stream);
int erc_mm_cross_grad(const float* dhi, int ldd, const float* h, int ldh, const int32_t* node_dlg,
                      const int32_t* node_off, int M, int N, int P, float* dCR, int planes, int64_t d_plane,
                      int64_t h_plane, void* stream);
/* GCNII layer tail (mmgcn_models.py:27-39,385-388): hd = dropout(relu(theta*G + (1-theta)((1-alpha) hi + alpha h0)));
 * hi == NULL: hd = dropout(relu(G)) (the input layer fcs[0]).  Backward: dG, dhi (written), dh0 (accumulated). */
int erc_gcnii_combine_fwd(const float* G, const float* hi, const float* h0, int64_t n, float theta, float alpha,
                          float drop_p, const uint64_t* rng_state, uint64_t rng_stream, float* hd, void* stream);
int erc_gcnii_combine_bwd(const float* d_hd, const float* hd, int64_t n, float theta, float alpha, float keep_scale,
                          int plain, float* dG, float* dhi, float* dh0, int F, int ld_d, void* stream);
/* One GCNII layer's dense part with its tail in the epilogue (mmgcn_models.py GraphConvolution.forward / GCNII.forward):
 *   hd = dropout(relu(theta * ([hi | h0] W) + (1 - theta) * ((1 - alpha) hi + alpha h0)))
 * hih0: rows [hi | h0] of pitch lda (both F wide); W [2F, F] stored [in, out]; dropout stream rng_stream of the counter
 * RNG (element index row * F + col), so the mask equals erc_gcnii_combine_fwd's.  One launch instead of two GEMMs and
 * the combine; in the backward dhi / dh0 are the halves of one [rows, 2F] buffer (pitch ld_d above) so that
 * [dhi | dh0] += dG W^T is one GEMM as well. */
int erc_gcnii_layer_fwd(const float* hih0, int lda, const float* W, int ldw, float theta, float alpha, float drop_p,
                        const uint64_t* rng_state, uint64_t rng_stream, float* hd, int ldo, int rows, int F, void* stream);
int erc_dropout_fwd(const float* x, int64_t n, float drop_p, const uint64_t* rng_state, uint64_t rng_stream, float* y,
                    void* stream);
/* FE[i, m*400 + c] = relu(dropout(cat[xd, h][(m,i), c])): regroup (mmgcn_models.py:570-576) + dropout_/ReLU
 * (mmgcn.py:119-120); backward scatters to d_xd / d_h. */
int erc_mm_regroup_fwd(const float* xd, const float* hl, int M, int N, float drop_p, const uint64_t* rng_state,
                       uint64_t rng_stream, float* FE, void* stream);
int erc_mm_regroup_bwd(const float* dFE, const float* FE, int M, int N, float keep_scale, float* d_xd, float* d_h,
                       void* stream);
/* y (+)= scale * x, optionally masked by mask != 0 */
int erc_axpy_mask(const float* x, const float* mask, int64_t n, float scale, int accumulate, float* y, void* stream);


/* Test support (not part of the data path): fills the LDS of every CU with NaN bit patterns, so that a persistent
 * kernel that reads LDS it did not initialise fails its parity test deterministically.  sink: one int32, may be NULL. */
int erc_test_poison_lds(int32_t* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ERCGRAFT_H */
