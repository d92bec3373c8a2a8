/*
 * leclip_hip.h - C ABI of the MI355X (gfx950) kernels behind the CLIP multi-label scoring path.
 *
 * The reference (JarvisUSTC/Language-Enhanced-CLIP-For-Multi-label-Image-Recognition) is 100 % Python on
 * PyTorch: it has no FFI for this path, so there is no reference-side native interface to copy.  Each entry
 * point below therefore names the reference *Python call* whose arithmetic it replaces (file:line under
 * /root/reference/project/my_code), and INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions (every function):
 *   - plain C, device pointers + explicit shapes / leading dimensions (in ELEMENTS) + dtype enums;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); functions only ENQUEUE work:
 *     no allocation, no synchronisation, no host<->device copies, safe under stream capture;
 *   - the caller owns every buffer including workspaces (sizes from the *_workspace_bytes helpers);
 *   - returns 0 on success, <0 on error (LECLIP_E_*); leclip_strerror() names the code and
 *     leclip_last_error() gives the per-thread detail string.  Never throws, never aborts;
 *   - re-entrant; calls on distinct streams may be issued from distinct host threads.
 *   - fp32 accumulation everywhere; LayerNorm / softmax statistics in fp32.
 */
#ifndef LECLIP_HIP_H
#define LECLIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LECLIP_ABI_VERSION 9

typedef enum { LECLIP_F32 = 0, LECLIP_F16 = 1, LECLIP_BF16 = 2 } leclip_dtype;
typedef enum { LECLIP_ACT_NONE = 0, LECLIP_ACT_QUICKGELU = 1 } leclip_act;
typedef enum { LECLIP_MASK_NONE = 0, LECLIP_MASK_CAUSAL = 1 } leclip_mask;

#define LECLIP_OK 0
#define LECLIP_E_INVALID (-1)      /* null pointer / non-positive size / bad enum */
#define LECLIP_E_UNSUPPORTED (-2)  /* shape or dtype combination this build has no kernel for */
#define LECLIP_E_LAUNCH (-3)       /* hipLaunchKernel reported an error */

int leclip_abi_version(void);
const char* leclip_strerror(int code);
const char* leclip_last_error(void);
/* Walk-order hint for the NEXT launches issued by the calling thread (thread-local; results never depend on it): 0 = tiles / (batch, head)
 * pairs in ascending row order, 1 = descending, -1 = the library's default (GEMMs ascending, the many-heads attention kernel descending).
 * A consumer that walks its rows in the order OPPOSITE to its producer's starts on the rows the producer wrote last - the part of a
 * 100 - 300 MB activation the 256 MiB memory-side cache still holds (hip/engine.py alternates the hint from launch to launch through a
 * residual block; profiles/r04_walk_order.txt).  The reference has no counterpart (its kernels are the vendor's, clip/model.py:213-228).
 * Returns the previous hint. */
int leclip_set_walk_order(int order);
/* GEMM kernel-family override for the NEXT launches issued by the calling thread (thread-local; results never depend on it - the three 16-bit
 * families feed v_mfma_f32_16x16x32 the same ascending K sequence and share one epilogue arithmetic, tests/test_gpu_parity.py holds them to bit
 * identity with this switch): 128 = gemm_tn_128x128x64, 256 = gemm_tn_256x256x64_pp, 384 = gemm_tn_384x256x32_pp (ABI 9, round 5), anything else =
 * the library's rate heuristic (tile count).  A family that cannot run a call (shape, epilogue) falls through to the next smaller one.  The
 * reference has no counterpart (its GEMMs are the vendor's, clip/model.py:213-228).  Returns the previous override (-1 = none). */
int leclip_set_gemm_family(int family);
/* Name of the device kernel family a GEMM call with these arguments dispatches to (for profiles/tests). */
const char* leclip_gemm_kernel_name(int64_t M, int N, int K, leclip_dtype ab_dtype);

/* LayerNorm over the last dimension, fp32 statistics, eps inside the rsqrt.
 * Replaces clip/model.py:193-199 (LayerNorm.forward: F.layer_norm(x.float(), ...).type(orig)).
 * y[r, :] = (x[r, :] - mean) * rsqrt(var + eps) * gamma + beta;  gamma/beta fp32 [dim].
 * dim must be a multiple of 64 and <= 4096. */
int leclip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                         int64_t rows, int dim, int64_t ldx, int64_t ldy, float eps,
                         leclip_dtype x_dtype, leclip_dtype y_dtype, void* stream);

/* Y[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[M,N]       (nn.Linear weight layout, K contiguous)
 * Replaces the F.linear calls of clip/model.py:213-217,223,226-227 (MHA in-proj / out-proj, mlp.c_fc + QuickGELU
 * (model.py:202-204: x*sigmoid(1.702x)), mlp.c_proj) with the bias, activation and residual add fused.
 * A and W share ab_dtype (F32: exact-fp32 MFMA path; F16/BF16: v_mfma_f32_16x16x32 in both kernel families, fp32 accumulate,
 * ascending-K order per output element - an element's bits do not depend on M or on the family that computed it).
 * bias: fp32 or NULL.  residual: res_dtype or NULL; may alias Y (in-place residual stream).
 * Constraints: N % 128 == 0 and K % 64 == 0 (F16/BF16); N % 64 == 0 and K % 32 == 0 (F32). */
int leclip_gemm_bias_act_res_fwd(const void* A, const void* W, const float* bias, const void* residual, void* Y,
                                 int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldr, int64_t ldy,
                                 leclip_act act, leclip_dtype ab_dtype, leclip_dtype res_dtype,
                                 leclip_dtype y_dtype, void* stream);

/* The same GEMM with LayerNorm folded around it (16-bit operands only).
 *  - ln_stats != NULL: A holds the UN-normalised rows x, W has gamma folded in (W'[n,k] = gamma[k] W[n,k]) and
 *      Y = act(rstd[m] * (x . W'^T - mean[m] * ln_colsum[n]) + bias[n]) + residual,
 *    with ln_stats [M][2] = (mean, rstd) per row, ln_colsum[n] = sum_k W'[n,k], bias[n] = sum_k beta[k] W[n,k] + b[n]:
 *    algebraically clip/model.py:193-199 followed by F.linear, without materialising (or 16-bit rounding) LN(x).
 *  - stats_out != NULL: additionally writes, per output row and 64-column block, (sum, sum of squared deviations from
 *    the block mean) of the rounded outputs to stats_out [N/64][M][2] - SLOT-major: the rows a wave finishes together are contiguous
 *    bytes of one slot, written with full-line stores (ABI 7; ABI 6 was row-major [M][N/64][2]); leclip_ln_stats_finalize_fwd merges the blocks
 *    (parallel-variance update) into (mean, rstd) for the next fused GEMM.  Plain stores in a fixed layout:
 *    deterministic, nothing to zero, no E[x^2] - mean^2 cancellation on rows with |mean| >> std. */
int leclip_gemm_ln_fused_fwd(const void* A, const void* W, const float* bias, const float* ln_stats,
                             const float* ln_colsum, const void* residual, void* Y, float* stats_out, int64_t M, int N,
                             int K, int64_t lda, int64_t ldw, int64_t ldr, int64_t ldy, leclip_act act,
                             leclip_dtype ab_dtype, leclip_dtype res_dtype, leclip_dtype y_dtype, void* stream);
/* The same fused-LayerNorm GEMM fed with the PRODUCER's block partials instead of finished statistics: ln_partials
 * [ln_slots][M][2] (ln_slots = K / 64, slot-major) as written by a stats_out epilogue or by leclip_patch_embed_ln_fwd.  The function
 * merges them (leclip_ln_stats_finalize_fwd's kernel) into ln_stats_ws [M][2] and runs the GEMM: one C call per consumer.
 * (Merging inside the 256x256 GEMM kernel was built and measured in round 2 and costs more than the launch it saves: the
 * specialised LayerNorm-epilogue kernels have no registers left for a row's partials - DESIGN.md section 6.)
 * Give ln_stats OR ln_partials. */
int leclip_gemm_ln_partials_fwd(const void* A, const void* W, const float* bias, const float* ln_stats, const float* ln_partials,
                                int ln_slots, float* ln_stats_ws, float ln_eps, const float* ln_colsum, const void* residual,
                                void* Y, float* stats_out, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldr,
                                int64_t ldy, leclip_act act, leclip_dtype ab_dtype, leclip_dtype res_dtype,
                                leclip_dtype y_dtype, void* stream);
/* (mean, rstd) per row: from the partial sums above (fixed summation order), or directly from rows of x. */
int leclip_ln_stats_finalize_fwd(const float* partials, float* stats, int64_t rows, int slots, int dim, float eps,
                                 void* stream);
/* Residual GEMM that also finishes the LayerNorm statistics of its output rows (ABI 9, round 5):
 *   Y = A W^T + bias + residual        (the residual-stream update of a block, clip/model.py:225-228: x + attention(...), x + mlp(...))
 *   stats[m] = (mean, rstd) of row m of Y over its N columns, eps inside the rsqrt (what the NEXT LayerNorm's fp32 pass computes, clip/model.py:193-199)
 * 16-bit operands.  partials_ws [N/64][M][2] fp32 receives the per-64-column block partials (sum, M2 about the block mean) as with
 * leclip_gemm_ln_partials_fwd's stats_out.  On gemm_tn_384x256x32_pp the last of a 384-row block's workgroups to finish merges the block's partials
 * inside the launch (tickets_ws: one uint32 per 384-row block, ZERO on entry and zero again when the launch has ended; null = never merge in
 * the launch); otherwise the merge kernel runs behind the GEMM on the same stream.  Same arithmetic, same bits, as leclip_ln_stats_finalize_fwd. */
int leclip_gemm_res_stats_fwd(const void* A, const void* W, const float* bias, const void* residual, void* Y, float* partials_ws, float* stats,
                              unsigned* tickets_ws, float ln_eps, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldr, int64_t ldy,
                              leclip_dtype ab_dtype, leclip_dtype res_dtype, leclip_dtype y_dtype, void* stream);
int leclip_row_stats_fwd(const void* x, float* stats, int64_t rows, int dim, int64_t ldx, float eps, leclip_dtype x_dtype,
                         void* stream);

/* Patch embedding: X[b, 0, :] = class_emb + pos[0];  X[b, 1+p, :] = conv_{k=s=P, no bias}(image)[b, :, p] + pos[1+p].
 * Replaces clip/model.py:260-264 (conv1, reshape/permute, class-token concat, positional add).
 * image [B,3,R,R] (img_dtype), conv weight pre-flattened Wp [width, Kp] in w_dtype with Kp = roundup(3*P*P, 64)
 * (zero padded), class_emb fp32 [width], pos fp32 [T, width], X [B, T, width] in x_dtype, T = (R/P)^2 + 1.
 * workspace: leclip_patch_embed_workspace_bytes() bytes (the im2col patch matrix in w_dtype). */
int64_t leclip_patch_embed_workspace_bytes(int64_t B, int R, int P, leclip_dtype w_dtype);
int leclip_patch_embed_fwd(const void* image, const void* Wp, const float* class_emb, const float* pos, void* X,
                           int64_t B, int R, int P, int width, leclip_dtype img_dtype, leclip_dtype w_dtype,
                           leclip_dtype x_dtype, void* workspace, void* stream);

/* The same patch embedding with ln_pre fused behind it: X[b, t, :] = ln_pre(embedding row) (clip/model.py:260-265).  The conv
 * GEMM writes a plain [B*G*G, width] matrix (fast 16-bit epilogue, no residual / row remap); one row-wise pass then does the
 * class-token concat, the positional add and the LayerNorm.  The positional add happens in fp32 before the LayerNorm (the
 * un-normalised embedding is never rounded with the positional term in it).  workspace: leclip_patch_embed_ln_workspace_bytes(). */
int64_t leclip_patch_embed_ln_workspace_bytes(int64_t B, int R, int P, int width, leclip_dtype w_dtype);
int leclip_patch_embed_ln_fwd(const void* image, const void* Wp, const float* class_emb, const float* pos, const float* gamma,
                              const float* beta, void* X, float* stats_out, int64_t B, int R, int P, int width,
                              leclip_dtype img_dtype, leclip_dtype w_dtype, leclip_dtype x_dtype, float eps, void* workspace,
                              void* stream);
/* stats_out (nullable) [width/64][B*T][2] (slot-major): (sum, M2 about the block mean) of the stored rows per 64-column block - the block
 * partials the first residual block's fused LayerNorm merges in place (leclip_gemm_ln_partials_fwd), so that no separate
 * row-statistics pass runs over the fresh residual stream. */

/* Multi-head self-attention core on a packed QKV buffer: out = softmax(q k^T * scale + mask) v per (batch, head).
 * Replaces the scaled-dot-product step inside nn.MultiheadAttention as called at clip/model.py:221-223
 * (mask: clip/model.py:364-370 additive -inf strictly above the diagonal for the text tower).
 * qkv [B*T, 3*heads*64] rows = tokens (batch-major), columns q|k|v each heads*64 (in_proj order); out [B*T, heads*64].
 * head_dim must be 64.  F16/BF16: T <= 224 single-pass MFMA kernels (pipelined across heads when there are many),
 * T <= 640 streaming (online-softmax) kernel; F32: T <= 588. */
int leclip_attention_fwd(const void* qkv, void* out, int64_t B, int T, int heads, int head_dim,
                         int64_t ld_qkv, int64_t ld_out, leclip_mask mask, float scale,
                         leclip_dtype dtype, void* stream);

/* The same for the FIRST q_rows query rows of every (batch, head) only (q_rows = 0 or T: all of them = leclip_attention_fwd); keys and
 * values are still the whole sequence.  Rows are computed in blocks of 32 (F32: singly) by the same code as the full call, so a
 * computed row has the same bits in both; out rows q_rows .. up to the end of the last block may be written (same values as the full
 * call would give them), the rest of out is not touched.  The reference has no such call: its last residual block
 * (clip/model.py:207-228 inside :266) computes every token although VisionTransformer.forward keeps the class token alone
 * (`ln_post(x[:, 0, :])`, :271) - the image engine uses this for that block. */
int leclip_attention_prefix_fwd(const void* qkv, void* out, int64_t B, int T, int heads, int head_dim,
                                int64_t ld_qkv, int64_t ld_out, leclip_mask mask, float scale, int q_rows,
                                leclip_dtype dtype, void* stream);

/* out[i, :] = LayerNorm(x[row_index[i], :]) @ proj        proj [dim, E] row-major (the reference's `x @ proj`)
 * Replaces clip/model.py:271-274 (ln_post(x[:,0,:]) @ proj) and clip/model.py:388-390 /
 * trainers/Caption_distill_double.py:90,100 (ln_final(x)[arange, argmax(tokens)] @ text_projection);
 * LayerNorm is row-wise, so gathering first is the same arithmetic.  out fp32 [n, E]. */
int leclip_gather_ln_proj_fwd(const void* x, const int64_t* row_index, const float* gamma, const float* beta,
                              const void* proj, float* out, int64_t n, int dim, int E, int64_t ldx, float eps,
                              leclip_dtype x_dtype, leclip_dtype proj_dtype, void* stream);

/* logits[b, c] = scale * <img[b]/||img[b]||, txt[c]/||txt[c]||>      (no epsilon in the norms)
 * Replaces clip/model.py:399-404 and trainers/Caption_distill_double.py:330-335.  fp32 in, fp32 out. */
int leclip_l2norm_logits_fwd(const float* img, const float* txt, float* logits, int64_t B, int C, int D,
                             float scale, void* stream);

/* x[n, t, :] = table[tokens[n, t], :] + pos[t, :]       (clip/model.py:380-382; Caption_distill_double.py:83-86)
 * table / pos fp32; x in x_dtype. */
int leclip_embed_tokens_fwd(const int64_t* tokens, const float* table, const float* pos, void* x,
                            int64_t n, int T, int dim, int64_t vocab, leclip_dtype x_dtype, void* stream);

/* x[c, :, :] = cat(prefix[c] (1 row), ctx (n_ctx rows; generic [n_ctx,dim] if ctx_per_class == 0 else [n_cls,n_ctx,dim]),
 *                  suffix[c] (T-1-n_ctx rows)) + pos      - PromptLearner.forward 'end' layout + TextEncoder's
 * positional add (trainers/Caption_distill_double.py:206-225, 86).  All inputs fp32; pos may be NULL
 * (plain PromptLearner.forward concat); x in x_dtype. */
int leclip_prompt_assemble_fwd(const float* prefix, const float* ctx, const float* suffix, const float* pos, void* x,
                               int64_t n_cls, int n_ctx, int T, int dim, int ctx_per_class,
                               leclip_dtype x_dtype, void* stream);

/* x[n, t, :] = in[n, t, :] + pos[t, :] with a cast to x_dtype: TextEncoder.forward's `prompts + positional_embedding`
 * on an already assembled fp32 prompt tensor (trainers/Caption_distill_double.py:86). */
int leclip_add_pos_fwd(const float* in, const float* pos, void* x, int64_t n, int T, int dim, leclip_dtype x_dtype,
                       void* stream);

/* Index of the maximum token id per row (first occurrence), = tokens.argmax(-1) (clip/model.py:390), plus the flat
 * row n*T + argmax used by leclip_gather_ln_proj_fwd. */
int leclip_eot_index_fwd(const int64_t* tokens, int64_t* eot, int64_t* flat_row, int64_t n, int T, void* stream);

/* The whole tail of the image branch in one launch, both contractions on the matrix cores:
 *   feat[b, :] = LayerNorm(x[b * row_stride : +dim]) @ proj          (clip/model.py:271-274: ln_post(x[:, 0, :]) @ proj)
 *   logits[b, c] = scale * <feat_b / |feat_b|, txt_c / |txt_c|>      (model.py:399-404; Caption_distill_double.py:330-335)
 * x in `dtype` (class-token row of image b at element offset b * row_stride), proj_t [E, dim] = proj^T in `dtype`
 * (K contiguous), txt [C, E] fp32 un-normalised text features.  feat [B, E] fp32 and logits [B, C] fp32 are each optional
 * (NULL), at least one must be given.  16-bit dtypes: v_mfma_f32_16x16x32 for the projection; fp32: the exact
 * v_mfma_f32_16x16x4_f32; the logit contraction always runs in exact fp32 MFMA on the unrounded features.
 * dim % 64 == 0, dim <= 1024, E % 16 == 0. */
int leclip_image_tail_fwd(const void* x, const float* gamma, const float* beta, const void* proj_t, const float* txt,
                          float* feat, float* logits, int64_t B, int64_t row_stride, int dim, int E, int C, float eps,
                          float scale, leclip_dtype dtype, void* stream);

/* Gradient of leclip_l2norm_logits_fwd w.r.t. the text features (image features are frozen, reference :762-765):
 * dtxt [C, D] from dlogits [B, C]; all fp32; D <= 1024. */
int leclip_l2norm_logits_bwd(const float* img, const float* txt, const float* dlogits, float* dtxt, int64_t B, int C, int D,
                             float scale, void* stream);

/* dst[i, :] = src[index[i], :] (n rows)  and  dst = 0; dst[index[i], :] = src[i, :] (n rows into dst_rows rows; indices
 * distinct).  The EOT-row gather / scatter around ln_final in the prompt-tuning backward (model.py:390). Rows of
 * dim elements, leading dimensions in elements, row bytes a multiple of 16. */
int leclip_gather_rows_fwd(const void* src, const int64_t* index, void* dst, int64_t n, int dim, int64_t ld_src, int64_t ld_dst,
                           leclip_dtype dtype, void* stream);
int leclip_scatter_rows_fwd(const void* src, const int64_t* index, void* dst, int64_t n, int64_t dst_rows, int dim, int64_t ld_src,
                            int64_t ld_dst, leclip_dtype dtype, void* stream);

/* Multi-crop test path: every window of every image through the reference's test transform, on the device.
 * Replaces `tfm(F.to_pil_image(block))` per window in DatasetWrapperWithBlock._transform_image
 * (dassl/data/data_manager.py:392-399, 417-425, ...) with tfm = Resize(S, bicubic) on the smaller edge + CenterCrop(S) +
 * ToTensor + Normalize (dassl/data/transforms/transforms.py:379-400).  The resize reproduces Pillow's 8-bit resampler
 * (libImaging/Resample.c: double-precision bicubic coefficients, 22-bit fixed point, horizontal pass then vertical pass)
 * bit for bit; the size rule is torchvision 0.12's.
 * src uint8 [B, 3, H, W] (planar); windows int32 [NW, 5] on the DEVICE = (y0, x0, rows, cols, pad_top): rows are counted on
 * the image after pad_top reflected rows were put on top (reflection also below the last row), columns are not padded;
 * the same NW windows are cut from each of the B images.  out [B, NW, 3, S, S] in out_dtype.  mean3 / std3: HOST pointers to
 * the three Normalize constants.  Windows are validated by the caller (rows / S and cols / S <= 15.5); the kernel clamps
 * every index, so a malformed window cannot fault. */
int leclip_crop_resize_fwd(const uint8_t* src, int64_t B, int H, int W, const int32_t* windows, int NW, void* out, int S,
                           const float* mean3, const float* std3, leclip_dtype out_dtype, void* stream);

/* ---- local ("dense") branch for a ViT.  The reference defines it for the ResNet only
 * (trainers/Caption_distill_double.py:401-472: per-position features from attnpool's v/c projections); for a ViT the
 * per-position features are the PATCH TOKENS of the last block taken through the same ln_post and proj as the class token.
 * x[r, :] /= |x[r, :]| in place (fp32): the `image_features / image_features.norm(dim=-1)` of :434 on [B*T, E] rows. */
int leclip_l2norm_rows_fwd(float* x, int64_t rows, int dim, int64_t ld, void* stream);
/* Spatial pooling of :447-462.  sim holds, per image, P rows (positions) of similarities against the "negative" prompts in
 * columns [0, C) and - when evidence_offset >= 0 - against the evidence prompts in columns [evidence_offset, +C); row p of
 * image b starts at sim + b * image_stride + p * ld.  Without evidence: prob = softmax over positions of scale * s,
 * out[b, c] = sum_p logit_scale * s * prob.  With evidence (winner-take-all): w = softmax over classes of
 * scale * s * (max_c s + 1), s <- s * w, prob = softmax over positions of scale * e.  All fp32; out [B, C]. */
int leclip_local_pool_fwd(const float* sim, float* out, int64_t B, int P, int C, int64_t ld, int64_t image_stride, int evidence_offset,
                          float spatial_scale, float logit_scale, void* stream);
/* The same pooling for the caption-as-image TRAINING branch (trainers/Caption_distill_double.py:473-513): the positions are the 77
 * token positions of a caption run through the text tower (`if_sequence=True`, :474) and `text_mask = (captions == 0) * -10000`
 * (:491) is added to both panels before anything else (:497-498, :505-506).  mask_tokens [B][mask_stride] int64 are the caption's
 * token ids (position p of image b is masked where the id is 0), or NULL for no mask. */
int leclip_local_pool_masked_fwd(const float* sim, const int64_t* mask_tokens, int64_t mask_stride, float* out, int64_t B, int P, int C,
                                 int64_t ld, int64_t image_stride, int evidence_offset, float spatial_scale, float logit_scale, void* stream);
/* Gradient of that pooling w.r.t. the similarity panels (the `ranking_loss(output_local, ...)` term of :806-808 on its way to
 * ctx_double / ctx_evidence): dout [B, C] -> dneg [B*P, C] (d / d s) and, with evidence, devi [B*P, C] (d / d e), contiguous.
 * torch's max(-1) in the winner-take-all weight (:509) hands its gradient to the arg-max class; so does this.  The panels of one
 * image must fit LDS (P * C * 12 bytes with evidence): the training branch pools 77 positions.
 * transposed_ld > 0: dneg / devi are written TRANSPOSED, [C][transposed_ld] with element (c, b * P + p) (transposed_ld >= B * P; pad
 * columns are not touched): the K-contiguous operand of the GEMM that contracts them with the position features (below). */
int leclip_local_pool_bwd(const float* sim, const int64_t* mask_tokens, int64_t mask_stride, const float* dout, float* dneg, float* devi,
                          int64_t B, int P, int C, int64_t ld, int64_t image_stride, int evidence_offset, float spatial_scale,
                          float logit_scale, int64_t transposed_ld, void* stream);
/* Pieces of the similarity GEMM's backward w.r.t. the text features (sim = normalise(positions) . normalise(text)^T, :495, :503):
 * dst [cols][ld_dst] = src [rows][ld_src]^T (fp32), and the backward of the row normalisation, dx = (dy - x_hat <x_hat, dy>) / |x|.
 * d text = l2norm_rows_bwd(text, gemm(dsim^T, positions_hat^T)): one exact-fp32 MFMA GEMM over all B * 77 rows instead of 80 workgroups
 * each streaming the whole position matrix. */
int leclip_transpose_f32_fwd(const float* src, float* dst, int64_t rows, int cols, int64_t ld_src, int64_t ld_dst, void* stream);
int leclip_l2norm_rows_bwd(const float* x, const float* dy, float* dx, int64_t rows, int dim, void* stream);

/* Caption-feature mixing of the test branch, trainers/Caption_distill_double.py:444-448: sim [B][ld_sim] = normalised global image
 * features . caption_text_feats^T (columns [0, N)), feats [N][E] the normalised caption features (generate_caption_text_features.py:82-88),
 * img [B][E] the normalised global image features:  out[b] = (img[b] + mean of the k rows of feats with the largest sim[b]) / 2
 * (the reference: topk = 10, `.topk(topk, -1)`, `.mean(1)`, `torch.cat([...], 1).mean(1)`).  Ties go to the lower index.  All fp32. */
int leclip_topk_mix_fwd(const float* sim, const float* feats, const float* img, float* out, int64_t B, int64_t N, int E, int k,
                        int64_t ld_sim, void* stream);

/* ---- score post-processing of the reference's test loop
 * Sliding-window aggregation, trainers/Caption_distill_double.py:654-660: window_logits [B, W, C] are the scores of the W
 * crops of each image; alpha = max_w, beta = min_w, s_ag = alpha > threshold ? alpha : beta, out = weight * s_ag + global
 * (reference: threshold 0.3, weight 1.4).  All fp32. */
int leclip_window_aggregate_fwd(const float* global_logits, const float* window_logits, float* out, int64_t B, int W, int C,
                                float threshold, float weight, void* stream);
/* Co-occurrence modulation, Caption_distill_double.py:614-618 (adjust_predictions) with the matrix of :632-634:
 * out = p + weight * (p @ Mn), Mn [C, C] = row-normalised(adj / nums[:, None]) built on the host (reference weight 0.5). */
int leclip_cooccurrence_adjust_fwd(const float* p, const float* Mn, float* out, int64_t B, int C, float weight, void* stream);

/* ---- native byte-level BPE tokenizer (host code).  Replaces SimpleTokenizer.encode / .bpe
 * (clip/simple_tokenizer.py:62-132) and clip.tokenize (clip/clip.py:185-221): whitespace collapse, lower-casing, the CLIP split
 * pattern with \p{L} / \p{N} / \s classified exactly as the `regex` module does, the GPT-2 byte alphabet, greedy lowest-rank
 * merging.  The merge table is the checkpoint-side file bpe_simple_vocab_16e6.txt.gz.  html entities and ftfy's repairs stay on
 * the caller's side: a text containing '&' (or U+03A3 / U+017F, whose case handling is context dependent) returns
 * LECLIP_E_UNSUPPORTED and the caller uses its Python path. */
void* leclip_bpe_open(const char* vocab_gz_path);                 /* NULL on failure: leclip_last_error() */
void leclip_bpe_close(void* handle);
int64_t leclip_bpe_vocab_size(void* handle);
/* ids of `utf8` without SOT / EOT; returns their number (writes at most max_ids) or a negative LECLIP_E_* */
int64_t leclip_bpe_encode(void* handle, const char* utf8, int64_t* ids, int64_t max_ids);
/* out [n, context_length] int64 = SOT + ids + EOT, zero padded; too long: error unless `truncate` (then the last kept id is EOT) */
int leclip_bpe_tokenize(void* handle, const char* const* texts, int64_t n, int context_length, int truncate, int64_t* out);

/* ---- backward of the text tower w.r.t. its ACTIVATIONS (prompt tuning: only the context vectors are trainable, reference
 * trainers/Caption_distill_double.py:762-765, 789-897).  dX of a linear layer is leclip_gemm_bias_act_res_fwd on a
 * transposed weight copy; the three kernels below cover the rest. */
/* dx = d LayerNorm(x)/dx applied to dy (gamma frozen), optionally + add (the residual branch's upstream gradient).
 * All tensors [rows, dim] contiguous in `dtype`; statistics recomputed in fp32. */
int leclip_layernorm_bwd(const void* dy, const void* x, const float* gamma, const void* add, void* dx, int64_t rows, int dim,
                         float eps, leclip_dtype dtype, void* stream);
/* QuickGELU as its own kernel for the training forward (the pre-activation must be kept) and its derivative:
 * out = pre * sigmoid(1.702 pre);   dpre = du * sigmoid(1.702 pre) * (1 + 1.702 pre (1 - sigmoid(1.702 pre))).  n % 4 == 0. */
int leclip_quickgelu_fwd(const void* pre, void* out, int64_t n, leclip_dtype dtype, void* stream);
int leclip_quickgelu_bwd(const void* pre, const void* du, void* dpre, int64_t n, leclip_dtype dtype, void* stream);
/* Gradient of leclip_attention_fwd w.r.t. the packed qkv (probabilities recomputed); T <= 104, head_dim 64. */
int leclip_attention_bwd(const void* qkv, const void* dout, void* dqkv, int64_t B, int T, int heads, int head_dim, int64_t ld_qkv,
                         int64_t ld_out, leclip_mask mask, float scale, leclip_dtype dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LECLIP_HIP_H */
