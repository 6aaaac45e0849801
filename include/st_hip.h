/* st_hip.h — C ABI of libst_hip.so: the MI355X (gfx950 / CDNA4) kernels of the GRPO hot path.
 *
 * The reference (hunarbatra/SpatialThinker) is 100 % Python and has no FFI of its own; every GPU
 * kernel it runs lives in a pip dependency (flash-attn, torch, vLLM — SURVEY.md §2.3).  Each entry
 * point below therefore cites the reference CALL SITE whose third-party kernel it replaces.
 *
 * Conventions
 *   - plain pointers + sizes; all pointers are DEVICE pointers unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work,
 *     never synchronise, never allocate; the caller owns every buffer;
 *   - bf16 tensors are `uint16_t` bit patterns; row-major; "ld*" = leading dimension in ELEMENTS;
 *   - return 0 on success, a positive hipError_t on a launch error, ST_EINVAL on bad arguments;
 *   - thread-compatible: concurrent calls must use distinct streams and distinct outputs.
 */
#ifndef ST_HIP_H
#define ST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ST_EINVAL (-22)
typedef void* st_stream_t;
typedef uint16_t st_bf16;

int st_version(void);                 /* ABI version, currently 1 */
const char* st_arch(void);            /* "gfx950" */

/* ---- profiling hooks (used by bench.py for the live `roofline.achieved` figure) ------------
 * When enabled for kernel-class `klass` (ST_K_*), every launch of that class is bracketed by
 * hipEventRecord on the launch stream; st_prof_read() synchronises those events and returns
 * the number of launches and their summed duration in milliseconds. */
enum { ST_K_GEMM = 0, ST_K_ATTN_FWD = 1, ST_K_ATTN_BWD = 2, ST_K_LOGPROB = 3, ST_K_ADAMW = 4,
       ST_K_RMSNORM = 5, ST_K_VIT_ATTN = 6, ST_K_DECODE_ATTN = 7, ST_K_GEMM_FP8 = 8, ST_K_VIT_WIN = 9 /* ViT window attention (attention_win.hip): units = bytes */, ST_K_COUNT = 10 };
int st_prof_enable(int klass, int max_events);
int st_prof_read(int klass, int* launches, double* total_ms, double* total_units);
int st_prof_disable(int klass);
/* per-launch view of the sampled launches (round 5: the per-shape table behind `roofline`, profiles/r05_gemm_shapes.json): duration in
 * ms, algorithmic units and the launcher's shape tag — GEMM class: form (0 NT, 1 NN, 2 TN, 3 SwiGLU) << 60 | epilogue flags (1 bias,
 * 2 residual, 4 fp32 out, 8 accumulate) << 56 | K << 36 | N << 18 | M; 0 for the other classes.  Fills up to `max` entries, returns the
 * sampled count in *n_out, resets the class like st_prof_read. */
int st_prof_read_events(int klass, int max, float* ms_out, double* units_out, unsigned long long* tags_out, int* n_out);
/* sample every stride-th launch of the class (default 1): max_events then spans stride * max_events launches of a long run */
int st_prof_set_stride(int klass, int stride);
/* algorithmic units (flops / bytes) of the NEXT launch of the class, for launchers whose work depends on device-side ranges
 * (attention: the host-side packing knows the (query, key) pair count); ignored while the class is disabled */
int st_prof_hint_units(int klass, double units);
/* launches of the class seen since st_prof_enable / the last st_prof_read (sampled or not; graph-captured ones excluded) */
int64_t st_prof_seen(int klass);

/* ---- fused log-prob (replaces flash_attn.ops.triton.cross_entropy, called from
 *      verl/utils/torch_functional.py:34-42 via verl/workers/actor/dp_actor.py:125-128) --------
 * logits (T, V) bf16 with row stride ldl; labels (T,) int64.  z = logits * inv_temperature
 * (dp_actor.py:126 `logits.div_(temperature)`), logp[t] = z[label] - logsumexp(z), fp32.
 * lse (T,) is saved for the backward.  Rows with label < 0 or >= V produce logp = 0. */
int st_logprob_fwd(const st_bf16* logits, int64_t ldl, const int64_t* labels, float inv_temperature,
                   float* logp, float* lse, int T, int V, st_stream_t stream);
/* in-place backward: logits[t, v] <- g[t] * inv_temperature * (onehot(label) - exp(z - lse)) (bf16). */
int st_logprob_bwd(st_bf16* logits, int64_t ldl, const int64_t* labels, const float* lse, const float* g,
                   float inv_temperature, int T, int V, st_stream_t stream);

/* ---- GRPO token-level loss (verl/trainer/core_algos.py:291-353 compute_policy_loss, :394-436
 *      compute_kl, verl/utils/torch_functional.py:69-71 masked_mean, composed as in
 *      verl/workers/actor/dp_actor.py:252-278) -------------------------------------------------
 * All inputs (n,) fp32 except mask (n,) int64 (the reference's response_mask dtype).  ref may be NULL
 * (no KL term).  kl_kind: 0 kl, 1 abs, 2 mse, 3 low_var_kl, 4 chi2.  Writes
 *   g[i]     = d( (pg_loss + kl_coef*kl_loss) / grad_accum ) / d logp[i]
 *   metrics  = { pg_loss(+kl term), pg_clipfrac_higher, pg_clipfrac_lower, ppo_kl, entropy_loss,
 *                kl_loss, sum(mask), 0 }  (8 floats, overwritten). One deterministic workgroup. */
int st_grpo_loss(const float* logp, const float* old_logp, const float* ref_logp, const float* adv,
                 const int64_t* mask, int n, double clip_low, double clip_high, double clip_dual,
                 int kl_kind, double kl_coef, double grad_accum, float* g, float* metrics, st_stream_t stream);

/* ---- critic (adv_estimator = gae; /root/reference/verl/workers/critic/dp_critic.py) -------------------------------------------------
 * st_value_head_fwd: v[t] = bf16(hn[t] . w + b) — the `score = nn.Linear(H, 1)` head whose `output.logits` dp_critic.py:104-118 reads.
 * st_value_head_bwd: dhn = dv (x) w (bf16), dw_accum += dv^T hn, db_accum += sum(dv) (fp32, deterministic).
 * st_value_loss: core_algos.py:356-392 compute_value_loss + its gradient: g = d(vf_loss / grad_accum)/dvpreds;
 *   metrics = [vf_loss, vf_clipfrac, masked_mean(vpreds), sum(mask)] (dp_critic.py:196-211). */
int st_value_head_fwd(const st_bf16* hn, int64_t ldh, const st_bf16* w, const st_bf16* bias, float* out, int T, int H, st_stream_t stream);
int st_value_head_bwd(const st_bf16* hn, int64_t ldh, const st_bf16* w, const float* dv, st_bf16* dhn, int64_t lddh, float* dw_accum,
                      float* db_accum, int T, int H, st_stream_t stream);
int st_value_loss(const float* vpreds, const float* returns, const float* values, const int64_t* mask, int n, double cliprange_value,
                  double grad_accum, float* g, float* metrics, st_stream_t stream);
/* ---- GRPO outcome advantage (verl/trainer/core_algos.py:137-175) ------------------------------
 * rewards (N, R) fp32, mask (N, R) int64, group (N,) int32 dense group index in [0, n_groups)
 * (the host maps uid strings to it).  score_i = sum_t rewards[i,t]; per group, members visited in
 * row order: mean = (sequential fp32 sum)/n, std = unbiased (Welford, fp64 accumulate, as torch's
 * CPU std); adv[i,t] = (score_i-mean)/(std+eps) * mask[i,t].  Groups with < 2 rows -> returns -1
 * through *status (device int, may be NULL). */
int st_grpo_advantage(const float* rewards, const int64_t* mask, const int32_t* group, int N, int R,
                      int n_groups, double eps, float* adv, float* scratch /* N + 2*n_groups floats */,
                      int32_t* status, st_stream_t stream);

/* ---- RMSNorm (HF Qwen2_5_VLRMSNorm modeling_qwen2_5_vl.py:65-79; 2 per LM layer + final + ViT) --
 * y = w * bf16(x * rsqrt(mean(x^2)+eps)) with the HF rounding points (normalised value rounded to
 * bf16 BEFORE the weight multiply).  rstd (T,) fp32 optional output for the backward. */
int st_rmsnorm_fwd(const st_bf16* x, int64_t ldx, const st_bf16* w, float eps, st_bf16* y, int64_t ldy,
                   float* rstd, int T, int H, st_stream_t stream);
/* dx = rstd * (w*dy - xhat * mean(w*dy*xhat)); dw_accum (H,) fp32 += sum_t dy*bf16(xhat) (fp32 atomics, may be
 * NULL).  dx may alias dy.  If dres != NULL, dx += dres (residual-stream gradient). */
int st_rmsnorm_bwd(const st_bf16* x, int64_t ldx, const st_bf16* w, const float* rstd, const st_bf16* dy,
                   int64_t lddy, const st_bf16* dres, int64_t lddres, st_bf16* dx, int64_t lddx,
                   float* dw_accum, int T, int H, st_stream_t stream);
/* The same backward in ONE pass over x / dy with a DETERMINISTIC dw (round 6): every workgroup leaves one fp32 partial row of dw in
 * `workspace` (st_rmsnorm_bwd_workspace_bytes(T, H) bytes, contents irrelevant), a second small kernel adds the rows to dw_accum in a fixed
 * order — no atomics, the same bits on every run; dx is bit-identical to st_rmsnorm_bwd's.  H <= 4096, 16-byte aligned operands.  Replaces
 * the autograd backward of HF Qwen2_5_VLRMSNorm under /root/reference verl/workers/actor/dp_actor.py:212-292 (loss.backward()). */
int64_t st_rmsnorm_bwd_workspace_bytes(int T, int H);
int st_rmsnorm_bwd_fused(const st_bf16* x, int64_t ldx, const st_bf16* w, const float* rstd, const st_bf16* dy, int64_t lddy,
                         const st_bf16* dres, int64_t lddres, st_bf16* dx, int64_t lddx, float* dw_accum, void* workspace,
                         int64_t workspace_bytes, int T, int H, st_stream_t stream);

/* ---- rotary embeddings ------------------------------------------------------------------------
 * M-RoPE (HF rotary_emb :525-538 + apply_multimodal_rotary_pos_emb :557-599, called from
 * verl/models/transformers/qwen2_vl.py:157-163).  The cos/sin table depends only on the positions,
 * so it is built ONCE per packed micro-batch and reused by every layer, forward and backward:
 *   pos (3, T) int32 (t,h,w rows); inv_freq (D/2,) fp32 = theta^(-2i/D) (host-computed as HF does);
 *   band i takes its angle from row 0 if i < s0, row 1 if i < s0+s1, else row 2 (s = 16,24,24);
 *   cos_out/sin_out (T, D/2) fp32, angle = fp32(pos) * inv_freq in fp32. */
int st_mrope_table(const int32_t* pos, const float* inv_freq, int T, int D, int s0, int s1, int s2, float* cos_out,
                   float* sin_out, st_stream_t stream);
/* Rotate, IN PLACE, the first n_rot_heads heads (head_dim D, D % 16 == 0) of every row of x (T, ld):
 *   y[i] = x[i]*c[i] - x[i+D/2]*s[i];  y[i+D/2] = x[i+D/2]*c[i] + x[i]*s[i]   (rotate_half, HF :153-157).
 * LM: x = qkv buffer, n_rot_heads = n_q + n_kv.  ViT (HF apply_rotary_pos_emb_vision :160-171): x =
 * (N, 3*heads*D) [q|k|v], n_rot_heads = 2*heads, tables (N, D/2) in the window-reordered row order.
 * inverse != 0 applies the transposed rotation (= backward). */
int st_rope_apply(st_bf16* x, int64_t ld, const float* cos_tab, const float* sin_tab, int T, int n_rot_heads, int D,
                  int inverse, st_stream_t stream);

/* ---- SwiGLU (HF Qwen2MLP :541-554 / Qwen2_5_VLMLP :82-96): gu (T, 2I) = [gate | up] ------------ */
int st_swiglu_fwd(const st_bf16* gu, int64_t ldgu, st_bf16* out, int64_t ldo, int T, int I, st_stream_t stream);
/* dgu = [dout*up*silu'(gate) | dout*silu(gate)], may alias gu. */
int st_swiglu_bwd(const st_bf16* gu, int64_t ldgu, const st_bf16* dout, int64_t lddo, st_bf16* dgu,
                  int64_t lddgu, int T, int I, st_stream_t stream);
/* st_swiglu_bwd that also writes m_out (T, I) = st_swiglu_fwd(gu), bit for bit, from the values it reads anyway (NULL: plain st_swiglu_bwd):
 * the backward that did not keep the activation needs it for the down projection's weight gradient. */
int st_swiglu_bwd_m(const st_bf16* gu, int64_t ldgu, const st_bf16* dout, int64_t lddo, st_bf16* dgu, int64_t lddgu, st_bf16* m_out,
                    int64_t ldm, int T, int I, st_stream_t stream);
/* exact (erf) GELU of the patch merger (HF :137-150), fwd and bwd (dx = dy * gelu'(x)). */
int st_gelu_fwd(const st_bf16* x, st_bf16* y, int64_t n, st_stream_t stream);
int st_gelu_bwd(const st_bf16* x, const st_bf16* dy, st_bf16* dx, int64_t n, st_stream_t stream);

/* ---- bf16 MFMA GEMM (replaces the cuBLAS/hipBLASLt calls behind every nn.Linear of the model) ---
 * C[M,N] = A[M,K] * B[N,K]^T  (both operands K-contiguous: y = x W^T with W as stored by HF).
 *   bias (N,) bf16 or NULL; residual (M,N) bf16 or NULL is added after bias;
 *   out_bf16 (M,N) bf16 or NULL; out_f32 (M,N) fp32 or NULL; if accumulate != 0 the fp32 output
 *   is C += result (gradient accumulation across micro-batches).  K % 64 == 0 required. */
int st_gemm_nt(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, const st_bf16* bias,
               const st_bf16* residual, int64_t ldr, st_bf16* out_bf16, float* out_f32, int64_t ldc,
               int accumulate, int M, int N, int K, st_stream_t stream);
/* Backward-pass operand layouts of the same GEMM tile (no transposed copies in HBM; the fragments of a contraction-major operand
 * come from hardware transpose reads of the LDS image):
 *   st_gemm_nn: out[M,N] = A[M,K] B[K,N]          — dX = dY W with W as stored (out_features x in_features): torch's dgrad GEMM behind
 *               the reference's loss.backward() (verl/workers/actor/dp_actor.py:277);
 *   st_gemm_tn: out_f32[M,N] (+)= A[K,M]^T B[K,N] — dW = dY^T X, accumulating into the flat fp32 gradient buffer (wgrad). */
int st_gemm_nn(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, st_bf16* out, int64_t ldc, int M, int N, int K,
               st_stream_t stream);
int st_gemm_tn(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, float* out_f32, int64_t ldc, int accumulate, int M, int N,
               int K, st_stream_t stream);

/* Optional workspace for st_gemm_nt / st_gemm_nt_variant (device memory, `bytes` long, 256 KiB per tail slice; 128 MiB covers
 * every split the library picks).  With it, a launch whose 256x256 tiles do not fill whole rounds of the device's CUs cuts its
 * last partial round into K-slices (fp32 partials in the workspace, summed in a fixed order by a second launch that also runs the
 * bias/residual/accumulate epilogue) — e.g. 574 tiles on 256 CUs finish in ~2.3 rounds instead of 3.  NULL / 0 switches it off.
 * The workspace is process-global: GEMMs that may use it must be issued on one stream at a time. */
int st_gemm_set_workspace(void* workspace, int64_t bytes);
/* Production tile of the training-shape launches of st_gemm_nt / st_gemm_swiglu: 23 = 256x256, 8 waves (mid-tile barrier schedule,
 * LDS-staged epilogue, tail split); 40 = 256x256, 4 waves x 128x128 with the hand-scheduled K loop (csrc/gemm_asm4.hip).  Process-wide;
 * also settable through the environment (ST_GEMM_VARIANT) before the library loads. */
int st_gemm_select(int variant);
/* Tuning/inspection entry: the same GEMM with an explicit tile variant (0: 128x128 2-stage, 1: 128x128 3-stage,
 * 2: 256x128 2-stage, 3: 256x128 3-stage, 4: 256x256 2-stage, 5: 128x256 3-stage, 6/7: 256x256 / 128x128 mid-tile barrier,
 * 8: 256x256 with 4 waves of 128x128 and a hand-written schedule, 9: the same tile, compiler schedule).  st_gemm_nt picks per shape. */
int st_gemm_nt_variant(int variant, const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, const st_bf16* bias,
                       const st_bf16* residual, int64_t ldr, st_bf16* out_bf16, float* out_f32, int64_t ldc, int accumulate,
                       int M, int N, int K, st_stream_t stream);
/* Decode-shaped variant (M <= 256, the rollout's one-token-per-sequence GEMMs: a weight stream bound by HBM,
 * 2*N*K bytes).  scratch (scratch_elems floats, contents irrelevant) receives the split-K partial slabs
 * [split][M][N], summed in a fixed order by a finish kernel; NULL disables split-K. */
int st_gemm_nt_skinny(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, const st_bf16* bias,
                      const st_bf16* residual, int64_t ldr, st_bf16* out_bf16, int64_t ldc, float* scratch,
                      int64_t scratch_elems, int M, int N, int K, st_stream_t stream);
/* gate/up projection + SwiGLU in one kernel for any M (training forward, no-grad log-prob passes, prefill):
 * m_out[M, I] = silu(A gate_w^T) * (A up_w^T); gu_out (NULL or [M, 2I]) also receives the bf16 gate|up projections the
 * backward needs (st_swiglu_bwd).  Bit-identical to st_gemm_nt + st_swiglu_fwd (HF Qwen2MLP.forward). */
int st_gemm_swiglu(const st_bf16* A, int64_t lda, const st_bf16* gate_up_w, int64_t ldb, st_bf16* gu_out, int64_t ldgu,
                   st_bf16* m_out, int64_t ldm, int M, int I, int K, st_stream_t stream);
/* Decode MLP up-projection with the SwiGLU fused into the GEMM epilogue (M <= 256):
 * out[M, I] = silu(A gate_w^T) * (A up_w^T) with gate_up_w = [gate_w ; up_w] (2I x K, the fused layout of ParamStore);
 * same bf16 rounding points as st_gemm_nt + st_swiglu_fwd (HF Qwen2MLP.forward), bit-identical to that pair. */
int st_gemm_swiglu_decode(const st_bf16* A, int64_t lda, const st_bf16* gate_up_w, int64_t ldb, st_bf16* out, int64_t ldc,
                          int M, int I, int K, st_stream_t stream);
/* tuning entry for st_gemm_swiglu_decode: the same operation on an explicit tile (variant 1..7, see gemm_tiles.hip) */
int st_gemm_swiglu_decode_variant(int variant, const st_bf16* A, int64_t lda, const st_bf16* gate_up_w, int64_t ldb, st_bf16* out,
                                  int64_t ldc, int M, int I, int K, st_stream_t stream);
/* tuning entry for the decode-shaped GEMM: explicit tile variant (10..18, see gemm_tiles.hip) and split-K count */
int st_gemm_nt_decode_variant(int variant, int splits, const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb,
                              const st_bf16* bias, const st_bf16* residual, int64_t ldr, st_bf16* out_bf16, int64_t ldc,
                              float* scratch, int64_t scratch_elems, int M, int N, int K, st_stream_t stream);
/* Inspection entries (no launch, host only): the (tile variant, split-K count) st_gemm_nt_skinny / st_gemm_nt_decode_slabs pick for a
 * decode-shaped GEMM of this size, and the tile id (st_gemm_swiglu_decode_variant numbering) st_gemm_swiglu_decode picks — the
 * production-shape parity tests assert that the 7B plans (128x128 / 256x128 / 256x256 tiles, the 256x160 SwiGLU tile, split-K >= 4)
 * are the ones under test.  Variant ids: gemm_tiles.hip (st_gemm_tile_decode). */
int st_gemm_decode_plan(int M, int N, int K, int64_t scratch_elems, int* variant_out, int* splits_out);
int st_gemm_swiglu_decode_plan(int M, int I, int* variant_out);
/* out (C, R) = in (R, C)^T, bf16 (operand re-layout for the backward GEMMs). */
int st_transpose(const st_bf16* in, int64_t ldin, st_bf16* out, int64_t ldout, int R, int C, st_stream_t stream);
/* column sums: out_f32 (C,) (+)= sum_r in[r, c]  (bias gradients). */
int st_colsum(const st_bf16* in, int64_t ldin, float* out_f32, int accumulate, int R, int C, st_stream_t stream);

/* ---- attention ---------------------------------------------------------------------------------
 * Causal varlen GQA flash attention over packed sequences (replaces flash_attn_varlen_func at
 * verl/models/transformers/flash_attention_utils.py:118-130; `repeat_kv` of qwen2_vl.py:165-166 is
 * never materialised).  q (T, n_q, D) / k,v (T, n_kv, D) given as base pointers + row strides (they
 * may live inside one qkv buffer).  cu_seqlens (n_seq+1,) int32.  D = 128.  out (T, n_q*D) bf16,
 * lse (n_q, T) fp32 (natural log, scaled scores); max_seqlen = longest sequence (sizes the grid).  causal=0 gives the bidirectional form used by
 * the ViT (HF :225-291) where D = 80 is supported as well; with D = 80, causal = 0, n_q == n_kv and max_seqlen <= 64 (the ViT's
 * windows) both directions run the one-(window, head)-per-workgroup kernels of csrc/attention_win.hip (forward bit-identical to the
 * generic kernel; backward in one launch, delta = sum_keys P dP; ST_VIT_WIN=0 in the environment keeps the generic kernels). */
int st_attn_fwd(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                const int32_t* cu_seqlens, int n_seq, int T, int n_q, int n_kv, int D, float scale, int causal,
                st_bf16* out, int64_t ldo, float* lse, int max_seqlen, st_stream_t stream);
/* backward: dq/dk/dv with the same layouts/strides as q/k/v (dk, dv summed over the q heads of a
 * group); delta (n_q, T) fp32 scratch.  D == 128 additionally needs `workspace` of at least
 * st_attn_bwd_workspace_bytes(T, n_q, D) bytes (per-query-head bf16 partials of dK/dV, reduced over each KV group in a
 * fixed order); D == 80 ignores it (may be NULL). */
int64_t st_attn_bwd_workspace_bytes(int T, int n_q, int D);
int st_attn_bwd(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                const st_bf16* out, int64_t ldo, const st_bf16* dout, int64_t lddo, const float* lse,
                const int32_t* cu_seqlens, int n_seq, int T, int n_q, int n_kv, int D, float scale, int causal,
                st_bf16* dq, int64_t lddq, st_bf16* dk, int64_t lddk, st_bf16* dv, int64_t lddv,
                float* delta, void* workspace, int64_t workspace_bytes, int max_seqlen, st_stream_t stream);
/* Shared-prefix ("segment") attention of a packed GRPO micro-batch (D = 128, causal).  The G rollouts of a prompt are packed
 * as [prompt][response_1]...[response_k]: segment s owns rows [seg_b[s], seg_e[s]); its queries see the prefix rows
 * [pre_b[s], pre_e[s]) entirely (empty for prompts / stand-alone sequences) and then their own rows causally — the prompt is
 * stored and computed ONCE per group where the reference runs it once per rollout (dp_actor.py:86-104 packs whole sequences).
 * Backward: dep_e[s] >= seg_e[s], rows [seg_e[s], dep_e[s]) = the queries outside the segment that see all its keys (the
 * group's response rows); T_valid = number of packed rows in use.  Same deterministic kernels and workspace as st_attn_bwd. */
int st_attn_fwd_seg(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                    const int32_t* seg_b, const int32_t* seg_e, const int32_t* pre_b, const int32_t* pre_e,
                    const st_bf16* k_pre, int64_t ldk_pre, const st_bf16* v_pre, int64_t ldv_pre /* NULL: the prefix rows index
                    k / v themselves; else they index these tensors (the prompt K/V cache left by the rollout prefill) */,
                    int n_seg, int T, int n_q, int n_kv, int D, float scale, st_bf16* out, int64_t ldo, float* lse, int max_seg,
                    st_stream_t stream);
int st_attn_bwd_seg(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                    const st_bf16* out, int64_t ldo, const st_bf16* dout, int64_t lddo, const float* lse,
                    const int32_t* seg_b, const int32_t* seg_e, const int32_t* pre_b, const int32_t* pre_e,
                    const int32_t* dep_e, int n_seg, int T, int T_valid, int n_q, int n_kv, int D, float scale,
                    st_bf16* dq, int64_t lddq, st_bf16* dk, int64_t lddk, st_bf16* dv, int64_t lddv, float* delta,
                    void* workspace, int64_t workspace_bytes, int max_seg, st_stream_t stream);
/* Attention over explicit row ranges (rollout decode; replaces vLLM paged attention,
 * verl/workers/rollout/vllm_rollout_spmd.py:141-143): sequence s has query rows [q_beg[s], q_end[s]) of q and key
 * rows [k_beg[s], k_end[s]) of k/v (device int32 arrays, read at run time so a captured hipGraph replays with
 * growing caches).  Non-causal.  Used twice per decode step: (i) per PROMPT — the G rollouts x (n_q/n_kv) heads
 * of a KV head form one query tile against the shared prompt keys (q laid out (B*g, n_kv*D), n_q = n_kv here);
 * (ii) per SAMPLE against its own generated keys.  Long prompts are split into key chunks (one "sequence" per
 * chunk, flash-decoding) whose partial outputs go to separate row slabs via o_beg.  Empty key ranges give out = 0,
 * lse = -inf. */
int st_attn_fwd_ranges(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                       const int32_t* q_beg, const int32_t* q_end, const int32_t* k_beg, const int32_t* k_end,
                       const int32_t* o_beg /* output row base per sequence, NULL = q_beg */,
                       int q_group /* g > 0: query row R = (sample R/g, head h*g + R%g) read in place from (B, n_q*D), 0: plain rows */,
                       int n_seq, int T_out, int n_q, int n_kv, int D, float scale, st_bf16* out, int64_t ldo,
                       float* lse /* (n_q, T_out) */, int max_q,
                       const int32_t* pre_beg, const int32_t* pre_end, const st_bf16* k_pre, int64_t ldk_pre,
                       const st_bf16* v_pre, int64_t ldv_pre /* optional (D = 128): a second key range per sequence, rows
                       [pre_beg, pre_end) of k_pre / v_pre, visited before the own range — lets ONE launch cover the shared-prompt
                       partials (keys in the prompt cache, own range empty) and the per-sample partials (prefix empty) */,
                       st_stream_t stream);
/* The per-SAMPLE partials of the decode step (use (ii) of st_attn_fwd_ranges; same reference call site, vllm_rollout_spmd.py:141-143) with
 * one WAVE per (item, head): items of at most 32 query rows against their own key range only (no prefix), D = 128, non-causal.  Each item
 * streams its keys in 32-key tiles through a private ring of `slots` (2..4) 16-KiB LDS slots — 32 KiB per workgroup at slots = 2, five
 * co-resident items per CU instead of the two 256-thread workgroups of attn_fwd128_kernel — with counted waits and no barriers
 * (round 6, csrc/attention_decode.hip).  Same outputs as st_attn_fwd_ranges for such items up to fp32 rounding of the online softmax
 * (32-key steps instead of 64): out rows [o_beg or q_beg, +Lq) of this head's 128 columns, lse (n_heads, T_out), lse = -inf and out
 * untouched for an empty key range. */
int st_attn_decode_rows(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                        const int32_t* q_beg, const int32_t* q_end, const int32_t* k_beg, const int32_t* k_end, const int32_t* o_beg,
                        int q_group, int n_items, int T_out, int n_heads, int D, float scale, st_bf16* out, int64_t ldo, float* lse,
                        int max_q, int slots, st_stream_t stream);
/* Which kernel st_attn_fwd_ranges uses for decode-shaped launches (n_q == n_kv, items of <= 64 query rows, D = 128): 1 = the persistent
 * one-workgroup-per-CU kernel (attn_decode128_kernel: the tiles of all of a workgroup's items stream through one 4-slot LDS ring, 3 tiles
 * in flight per CU), 0 = one workgroup per item (attn_fwd128_kernel<false>; default — measured faster, see attention.hip — and the
 * path of wider items).  Same arithmetic: bit-identical partials (tests/test_gpu_kernels.py).  ST_DECODE_ATTN=persistent sets the initial value to 1. */
int st_decode_attn_select(int persistent);   /* 2 = the persistent kernel with TWO workgroups per CU (2-slot rings, 80 KiB of LDS each) */
int64_t st_switch_value(int which);         /* 0: training GEMM tile id (40), 1: nt decode weight stream (1), 2: decode attention kernel (0), 3: nt K/V copies in the decode attention (1); -1 for an unknown id */
int64_t st_decode_attn_selected(void);       /* the current choice (0 unless ST_DECODE_ATTN / st_decode_attn_select say otherwise; pinned by tests/test_layout.py) */
/* Flash-decoding merge of n_parts partial attentions over disjoint key sets: parts (n_parts*rows, heads*D) bf16 with
 * their lse (heads, n_parts*rows) -> out (rows, heads*D); partials with lse = -inf (empty key range) are skipped. */
int st_attn_merge(const st_bf16* parts, int64_t ldp, const float* lse, int n_parts, st_bf16* out, int64_t ldo, int rows,
                  int heads, int D, int q_group /* g > 0: write row r, head h to out[r/g][(h*g + r%g)*D] */, st_stream_t stream);
/* ---- fused decode epilogues (rollout decode step; replace finish + RMSNorm / finish + RoPE + KV-append launch chains) ----
 * st_gemm_nt_decode_slabs: decode-shaped GEMM (M <= 256) that stops at the fp32 split-K slabs [splits][M][N] in `scratch`
 * (*splits_out >= 1 slabs; host int written synchronously).
 * st_decode_finish_norm:  x_out = bf16(sum slabs + residual);  h_out = RMSNorm(x_out) * norm_w  (norm_w NULL: finish only).
 * st_decode_finish_qkv:   row = bf16(sum slabs + bias) of the fused q|k|v projection; M-RoPE on the q and k heads
 *                         (cos/sin (B, D/2) fp32 as st_mrope_table makes them); q -> q_out[b], k, v -> kg/vg[b, gen_len[b], :].
 * Bit-identical to st_gemm_nt_skinny -> st_rmsnorm_fwd and st_gemm_nt_skinny -> st_rope_apply -> st_kv_append. */
int st_gemm_nt_decode_slabs(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, float* scratch, int64_t scratch_elems,
                            int M, int N, int K, int* splits_out, st_stream_t stream);
int st_decode_finish_norm(const float* slabs, int splits, const st_bf16* residual, int64_t ldr, st_bf16* x_out, int64_t ldx,
                          const st_bf16* norm_w, float eps, st_bf16* h_out, int64_t ldh, int M, int N, st_stream_t stream);
int st_decode_finish_qkv(const float* slabs, int splits, const st_bf16* bias, const float* cos_tab, const float* sin_tab,
                         st_bf16* q_out, int64_t ldq, st_bf16* kg, st_bf16* vg, int64_t gen_stride, const int32_t* gen_len,
                         const int32_t* row_map /* NULL or (B,): cache slot (sample id) of row b */, int B, int M, int n_q,
                         int n_kv, int D, st_stream_t stream);
/* KV-cache append: for each sample b (active[b] != 0 or active NULL) copy the K and V column slices of qkv row b
 * into kg/vg[b, gen_len[b], :width]; if increment, gen_len[b] += 1 afterwards. */
int st_kv_append(const st_bf16* qkv, int64_t ld, int col_k, int col_v, int width, st_bf16* kg, st_bf16* vg,
                 int64_t gen_stride, int32_t* gen_len, const int32_t* active, int B, int increment, st_stream_t stream);

/* ---- block-scaled fp8 GEMM (BASELINE.json config #5: "fp8 MFMA"; the reference has no fp8 path — its GEMMs are torch/cuBLAS bf16
 *      behind verl/workers/actor/dp_actor.py:118-124 — this is the MI355X-native fast path for the same Linear layers) -------------
 * MX-fp8: OCP e4m3fn elements + one e8m0 (power-of-two) scale per 32 consecutive k.
 * st_mxfp8_quantize: x (R, K) bf16 -> q (R, K) bytes and scales[K/128][scale_rows] dwords (byte j of scales[kt][r] = scale of
 *   k-block 4*kt + j of row r; scale = 2^(byte - 127), shared exponent = floor(log2(max|x|)) - 8, elements RNE, saturating at 448).
 *   K % 128 == 0, scale_rows >= R and a multiple of 4.
 * st_gemm_mxfp8_nt: out[M,N] (bf16) = dequant(A)[M,K] dequant(B)[N,K]^T (+bias)(+residual), fp32 accumulation on the block-scaled
 *   MFMAs (the only fp8 MFMAs that run at twice the bf16 rate on gfx950): the 4-wave hand-scheduled 256x256 tile on
 *   v_mfma_scale_f32_32x32x64_f8f6f4 (gemm_mx4.hip; default) or the 8-wave tile on v_mfma_scale_f32_16x16x128_f8f6f4 (gemm_fp8.hip).
 * st_gemm_mxfp8_select: waves = 4 | 8 picks the tile (A/B runs, tests; ST_FP8_TILE=8 sets the initial value);
 *   50 / 52 / 56 = the 4-wave tile on its K-tile schedules 0 / 2 / 6 (same results; tools/mx4_ksweep.py); anything else is refused. */
int st_mxfp8_quantize(const st_bf16* x, int64_t ldx, uint8_t* q, int64_t ldq, uint32_t* scales, int64_t scale_rows, int R, int K,
                      st_stream_t stream);
/* st_mxfp8_quantize_t: the quantisation of x^T without materialising it — x (R, C) bf16 -> qT (C, R) bytes + scales[R/128][scale_rows >= C]
 *   (MX blocks of 32 consecutive ROWS of x): the operands of the weight-gradient products dW = dY^T X.  R % 128 == 0, C % 8 == 0.
 * st_gemm_mxfp8_nt_f32: out[M,N] (fp32) = or += dequant(A)[M,K] dequant(B)[N,K]^T on the 4-wave tile. */
int st_mxfp8_quantize_t(const st_bf16* x, int64_t ldx, uint8_t* qT, int64_t ldq, uint32_t* scales, int64_t scale_rows, int R, int C,
                        st_stream_t stream);
/* st_mxfp8_quantize + st_mxfp8_quantize_t of the same x in ONE pass over it (a gradient that feeds both an fp8 input-gradient and an fp8
 * weight-gradient GEMM): q / scales as st_mxfp8_quantize, qT / scales_t as st_mxfp8_quantize_t, bit for bit.  R % 128 == 0, C % 128 == 0. */
int st_mxfp8_quantize_both(const st_bf16* x, int64_t ldx, uint8_t* q, int64_t ldq, uint32_t* scales, int64_t scale_rows, uint8_t* qT, int64_t ldqT,
                           uint32_t* scales_t, int64_t scale_t_rows, int R, int C, st_stream_t stream);
int st_gemm_mxfp8_nt_f32(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb,
                         const uint32_t* SB, int64_t sb_rows, float* out, int64_t ldc, int accumulate, int M, int N, int K, st_stream_t stream);
int st_gemm_mxfp8_select(int waves);
int st_gemm_mxfp8_nt(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb,
                     const uint32_t* SB, int64_t sb_rows, const st_bf16* bias, const st_bf16* residual, int64_t ldr, st_bf16* out,
                     int64_t ldc, int M, int N, int K, st_stream_t stream);
/* out[M,N] = bf16(silu(gate)) * up with [gate | up] = dequant(A)[M,K] dequant(B)[2N,K]^T (B: the N gate rows, then the N up rows):
 * the MLP's first product with the SwiGLU in the epilogue of the 4-wave fp8 tile (no-grad passes; roundings of st_gemm_swiglu). */
int st_gemm_mxfp8_swiglu(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb,
                         const uint32_t* SB, int64_t sb_rows, st_bf16* out, int64_t ldc, int M, int N, int K, st_stream_t stream);
/* The same product with the result leaving as MX-fp8 (the operand of the down projection): q (M, N) e4m3 bytes + sq[N/128][sq_rows] scale
 * dwords = st_mxfp8_quantize of st_gemm_mxfp8_swiglu's bf16 result, bit for bit; the bf16 activation is never stored.  N % 128 == 0. */
int st_gemm_mxfp8_swiglu_q(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb,
                           const uint32_t* SB, int64_t sb_rows, uint8_t* q, int64_t ldq, uint32_t* sq, int64_t sq_rows, int M, int N, int K,
                           st_stream_t stream);
/* Producers that emit the MX-fp8 operand of the next GEMM in the same pass (bit-identical to the bf16 op followed by st_mxfp8_quantize):
 * st_rmsnorm_mxfp8: RMSNorm forward (HF rounding points, modeling_qwen2_5_vl.py:74-79); y optional (NULL: the bf16 result is not kept),
 *   rstd optional.  H % 128 == 0.
 * st_swiglu_mxfp8: SwiGLU forward on gu = [gate | up] (T, 2I); out optional.  I % 128 == 0. */
int st_rmsnorm_mxfp8(const st_bf16* x, int64_t ldx, const st_bf16* w, float eps, st_bf16* y, int64_t ldy, uint8_t* q, int64_t ldq,
                     uint32_t* scales, int64_t scale_rows, float* rstd, int T, int H, st_stream_t stream);
int st_swiglu_mxfp8(const st_bf16* gu, int64_t ldgu, st_bf16* out, int64_t ldo, uint8_t* q, int64_t ldq, uint32_t* scales, int64_t scale_rows,
                    int T, int I, st_stream_t stream);

/* ---- optimizer: AnyPrecisionAdamW with bf16 states + Kahan compensation, one fused pass
 *      (verl/utils/torch_functional.py:253-329; ~10 eager passes in the reference) ---------------
 * p, m, v, c (n,) bf16 in place; grad fp32 (the fp32 accumulation buffer; rounded to bf16 first, the
 * dtype AnyPrecisionAdamW sees) scaled by grad_scale (clip coefficient, device scalar pointer or NULL).
 * lr..weight_decay are the python (double) hyper-parameters; 1-lr*wd, 1-b1, 1-b2 are formed in double and
 * cast once to fp32, as torch does with python scalars.  step_size = lr/(1-b1^t), denom_corr = sqrt(1-b2^t)
 * are host-computed in fp32 (torch computes them on a float32 0-d `step` tensor). */
int st_adamw_kahan_step(st_bf16* p, const float* grad, st_bf16* m, st_bf16* v, st_bf16* c, int64_t n,
                        double lr, double beta1, double beta2, double eps, double weight_decay, float step_size,
                        float denom_corr, const float* grad_scale, st_stream_t stream);
/* torch.optim.AdamW(fused=True) semantics on bf16 p / exp_avg m / exp_avg_sq v, no compensation buffer: the optimizer the
 * reference builds for worker.actor.optim.strategy=adamw (verl/workers/fsdp_workers.py:284-291).  grad fp32, rounded to bf16
 * (the parameter dtype torch's fused kernel requires of a gradient) after scaling by grad_scale.  bias_correction1 = 1-b1^t,
 * bias_correction2_sqrt = sqrt(1-b2^t), host-computed. */
int st_adamw_step(st_bf16* p, const float* grad, st_bf16* m, st_bf16* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, float bias_correction1, float bias_correction2_sqrt,
                  const float* grad_scale, st_stream_t stream);
/* torch.optim.AdamW(fused=True) on FP32 master parameters with fp32 exp_avg / exp_avg_sq — the reference's default actor
 * (worker.actor.fsdp.torch_dtype unset: fp32 parameters under MixedPrecision(param_dtype=bf16), verl/workers/fsdp_workers.py:186-189,
 * 238-243, 284-291).  p_bf16 receives the bf16 rounding of the updated master (the working copy every kernel computes with = what
 * FSDP's next all-gather in param_dtype yields).  grad fp32, scaled by grad_scale[0] (the clip coefficient). */
int st_adamw_master_step(float* master, st_bf16* p_bf16, const float* grad, float* m, float* v, int64_t n, double lr, double beta1,
                         double beta2, double eps, double weight_decay, float bias_correction1, float bias_correction2_sqrt,
                         const float* grad_scale, st_stream_t stream);
/* sum of squares of an fp32 buffer into out[0] (+= if accumulate) — global grad-norm for clipping
 * (verl/workers/actor/dp_actor.py:155-167). Deterministic two-stage reduction, scratch >= 1024 floats. */
int st_sumsq_f32(const float* x, int64_t n, float* scratch, float* out, int accumulate, st_stream_t stream);

/* ---- gather / scatter plumbing of the packed layout (flash_attn.bert_padding at
 *      verl/workers/actor/dp_actor.py:86-101,136-138; embed + masked_scatter HF :1205-1215) --------*/
int st_embed_gather(const st_bf16* table, int64_t ldt, const int32_t* ids, st_bf16* out, int64_t ldo, int T,
                    int H, st_stream_t stream);
int st_rows_gather(const st_bf16* src, int64_t lds, const int32_t* rows, st_bf16* dst, int64_t ldd, int n_rows,
                   int H, st_stream_t stream);                     /* dst[i] = src[rows[i]] */
int st_rows_scatter(const st_bf16* src, int64_t lds, const int32_t* rows, st_bf16* dst, int64_t ldd, int n_rows,
                    int H, int add, st_stream_t stream);           /* dst[rows[i]] (+)= src[i], rows unique */
/* dst[u] = bf16(sum over r < k of src[idx[u*k + r]]) accumulated in fp32 in the fixed order r = 0..k-1 (entries < 0 are
 * skipped): backward of a row gather with repeated sources.  Used for the image features that the G rollouts of one prompt share
 * (the reference runs the vision tower once per SEQUENCE, dp_actor.py:78-83; here once per distinct image of a micro-batch). */
int st_rows_gather_sum(const st_bf16* src, int64_t lds, const int32_t* idx, int k, st_bf16* dst, int64_t ldd, int n_out, int H,
                       st_stream_t stream);
int st_embed_grad(const st_bf16* dx, int64_t ldx, const int32_t* ids, float* dtable, int64_t ldt, int T, int H,
                  st_stream_t stream);                             /* dtable[ids[t]] += dx[t] (fp32 atomics) */
int st_cast_pad_f32_bf16(const float* in, int64_t ldin, st_bf16* out, int64_t ldout, int R, int C_in, int C_out,
                         st_stream_t stream);                      /* pixel_values fp32 -> bf16, zero-padded cols */
int st_add_bf16(const st_bf16* a, const st_bf16* b, st_bf16* out, int64_t n, st_stream_t stream);

/* ---- sampling (vLLM sampler semantics of rollout/config.py:25-32: temperature, top_k = -1, top_p = 1) ------
 * logits (B, V) bf16 -> one token per row.  temperature > 0: exact multinomial sampling of softmax(z/T) by the
 * Gumbel-max trick with a counter-based RNG keyed by (seed, step, row, index) — reproducible, no state;
 * temperature == 0: argmax.  forced (B,) int32 or NULL: entries >= 0 override the sampled token (EOS forcing
 * of the synthetic benchmark / max-length handling).  top_k > 0 / top_p < 1: vLLM order (top-k mask, then top-p on the rest);
 * the cut is exact over the 65536 possible bf16 values (two-level histogram), ties with the threshold value are kept; needs
 * scratch of B*33 floats. */
int st_sample(const st_bf16* logits, int64_t ldl, int B, int V, float temperature, int top_k, float top_p,
              uint64_t seed, uint64_t step, const int64_t* step_dev, const int32_t* forced,
              const int32_t* row_ids /* NULL or (B,): identity of each row for the RNG key (stable under batch compaction) */,
              const int32_t* row_steps /* NULL or (B,): per-row step overriding step/step_dev (rows at different response indices) */,
              int32_t* out_ids, float* scratch, st_stream_t stream);

/* ---- rollout decode: the bookkeeping between two decode forwards in ONE launch (vLLM's sampler output processing + scheduler state
 *      behind verl/workers/rollout/vllm_rollout_spmd.py:141-147; here ~25 one-element-per-row launches of the captured iteration) ------
 * st_sample_partials: st_sample without its last stage — the 16 partial (value, index) pairs per row stay in scratch (B*33 floats).
 * st_decode_step, per row b (one workgroup): token = argmax of the partials (forced_token where forced_len[b] == gen_len[b] + 1);
 *   if active[b]: out_tokens[b, min(gen_len[b], R-1)] = token;  active[b] &= !(gen_len[b] + 1 >= R || token in eos_ids (unless ignore_eos));
 *   tok_out[b] = token;  slot_out[b] = min(gen_len[b], R-1) (cache slot of this token's K/V);  gen_len[b] += 1 while the row was active
 *   on entry (round 5: a finished row keeps its response length);
 *   ke_gen[c*B + b] = clamp(k_base[b] + slot + 1, kb_gen[c*B + b], kb_gen[c*B + b] + chunk_keys) for the n_chunks generated-key chunks
 *   of a row that was active on entry, kb_gen[c*B + b] (EMPTY range) otherwise — a finished row stays in the GEMM tiles until the phase
 *   is re-batched, but the attention launch streams no generated K/V for it any more (round 5);
 *   cos_out/sin_out[b, :] = M-RoPE table row of pos[:, b] (as st_mrope_table), then pos[:, b] += 1;  x_out[b, :H] = embed[token, :H].
 * Round 4, optional (lse_partials / logp_out non-NULL): the rollout's OWN log-probabilities.  st_sample_partials additionally leaves, per row
 *   and split, (max, sum exp(z - max)) of the unfiltered z = logit / T in lse_partials (B*32 floats); st_decode_step then writes
 *   logp_out[b, slot] = logits[b, token] / T - logsumexp(z) for live rows — log pi_old(token) of the policy that sampled it, in the same pass
 *   over the logits (what vLLM's `logprobs` would return; the reference recomputes it with a second forward, fsdp_workers.py compute_log_probs). */
int st_sample_partials(const st_bf16* logits, int64_t ldl, int B, int V, float temperature, int top_k, float top_p, uint64_t seed,
                       uint64_t step, const int64_t* step_dev, const int32_t* row_ids, const int32_t* row_steps, float* scratch,
                       float* lse_partials, st_stream_t stream);
int st_decode_step(const float* sample_scratch, const int32_t* forced_len, int32_t forced_token, const int64_t* eos_ids, int n_eos,
                   int ignore_eos, int32_t* gen_len, int32_t* active, int64_t* out_tokens, int R, int32_t* tok_out, int32_t* slot_out,
                   const int32_t* k_base, const int32_t* kb_gen, int32_t* ke_gen, int n_chunks, int chunk_keys, int32_t* pos,
                   const float* inv_freq, int D, int s0, int s1, int s2, float* cos_out, float* sin_out, const st_bf16* embed,
                   int64_t ld_embed, st_bf16* x_out, int64_t ldx, int H, int B, const float* lse_partials, const st_bf16* logits, int64_t ldl,
                   float temperature, float* logp_out, st_stream_t stream);

/* ---- CU-partitioned streams (round 5: the decode tail of the rollout runs beside the old-policy log-prob pass of the finished samples;
 *      reference phases verl/trainer/ray_trainer.py:585-640 run one after the other) -----------------------------------------------
 * st_stream_create_cu_range: a HIP stream whose kernels run only on compute units [first_cu, first_cu + n_cus) of the chip's CU-mask
 * bit order (hipExtStreamCreateWithCUMask).  On MI355X bit i is slot i / 8 of XCD i % 8 (tools/probes/cu_mask_probe.hip), so a range
 * whose bounds are multiples of 8 takes the same number of CUs from every XCD.  EXPERIMENTAL, and measured as a NET LOSS for the use it was
 * built for: a register-only MFMA kernel and a streaming kernel on disjoint ranges keep their stand-alone rates (probe: 64 CUs stream
 * 3.4 TB/s next to 192 CUs at 1.64 PF/s), but real passes share every XCD's L2 and the fabric — the old-policy pass beside the decode tail
 * ran gen + old 12.35-12.40 s against 11.99 s serial (profiles/r05_notes.md §2), so bench.py keeps it off by default.  The handle is a
 * hipStream_t for every st_* entry's `stream` argument.  st_stream_destroy releases it (the stream must be idle). */
int st_stream_create_cu_range(int first_cu, int n_cus, st_stream_t* stream_out);
int st_stream_destroy(st_stream_t stream);

/* ---- measurement aid (round 6; no reference counterpart: the reference reports `perf/mfu_actor` against a fixed peak,
 *      verl/utils/flops_counter.py:24-50, and cannot say what clock the chip ran at) -------------------------------------------------
 * st_clock_probe: n_blocks one-wave workgroups (block b lands on XCD b % 8) each spin for spin_us microseconds of the constant 100-MHz
 * reference counter (s_memrealtime) and write {shader cycles (s_memtime), reference ticks} to out[2 b], out[2 b + 1]: shader clock in MHz =
 * 100 * out[2 b] / out[2 b + 1] — the clock the power management grants while whatever else is running runs.  Launched on a side stream
 * beside the timed region by bench.py (`roofline.clock_mhz`); 8 waves for spin_us, no memory traffic besides the 16 result bytes. */
int st_clock_probe(uint64_t* out, int n_blocks, int spin_us, st_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ST_HIP_H */
