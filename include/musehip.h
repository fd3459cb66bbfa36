/* musehip.h -- C ABI of libmusehip.so: the MI355X (gfx950) kernels behind the MuseDiffusion
 * hot path (GaussianDiffusion sampling / training losses over the TransformerNetModel denoiser).
 *
 * The reference (YAIxPOZAlabs/MuseDiffusion) has no native code and no FFI: its boundary for this
 * path is the Python surface MuseDiffusion.models.{network,diffusion,rounding}.  Each entry point
 * below replaces one torch-op group of that surface; the comment on each names the reference
 * file:line it stands in for.  The Python host (musediffusion_amd/) binds these with ctypes; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless named host_*.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), allocates nothing,
 *     never synchronises, and is hipGraph-capture safe.
 *   - return 0 on success, negative mh_status otherwise; mh_last_error() gives the message
 *     (thread-local).
 *   - `dtype` selects the activation/weight element type of the compute path:
 *     MH_F32 (fp32 storage, fp32-input MFMA / VALU: the parity mode) or MH_BF16 (bf16 storage,
 *     bf16 MFMA with fp32 accumulation: the throughput mode).  Latents, biases, LayerNorm
 *     parameters, embeddings used for rounding / logits and all diffusion arithmetic are fp32 in
 *     both modes.  MH_BF16X3 / MH_F16X3 (the denoiser forward only): the mode between the two - see "split precision" below.
 *   - "ld*" arguments are leading dimensions in ELEMENTS.
 */
#ifndef MUSEHIP_H
#define MUSEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mh_stream_t; /* hipStream_t */

enum mh_status { MH_OK = 0, MH_ERR_INVALID = -1, MH_ERR_HIP = -2, MH_ERR_UNSUPPORTED = -3 };
enum mh_dtype { MH_F32 = 0, MH_BF16 = 1,
                /* split precision (csrc/split.hip): every value as hi + lo 16-bit parts, three matrix-pipe products per reference product,
                 * fp32 accumulation - the denoiser forward only (mh_denoiser_*, mh_split_*), 4 bytes per element like MH_F32 */
                MH_BF16X3 = 2, MH_F16X3 = 3 };
enum mh_act { MH_ACT_NONE = 0, MH_ACT_TANH = 1, MH_ACT_GELU_ERF = 2, MH_ACT_SILU = 3,
              MH_ACT_DERIV = 4 /* mh_gemm_act_grad only: `pre` holds act'(pre) already (mh_gemm_bias_act_dact) */ };

const char* mh_last_error(void);
int mh_abi_version(void);
/* name of the device the library sees as device `ordinal` (diagnostics) */
int mh_device_name(int ordinal, char* buf, int buflen);

/* ------------------------------------------------------------------ layout / packing helpers */

/* out[r, c] = (T) in[r, c] for r < rows, c < cols; zero elsewhere in the [rows_out, ld_out] image.
 * Used to pack nn.Linear weights ([out,in] row-major, y = x W^T + b) into the weight
 * arena with K padded to the GEMM tile depth, and to cast the fp32 latent to the compute dtype
 * (models/network.py:141-144 input of input_up_proj). */
int mh_cast_pad(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int64_t cols,
                int64_t rows_out, int dtype, mh_stream_t stream);

/* out[r, c] = (float) in[r, c]: compute dtype -> fp32 (network.py:157 `h.type(x.dtype)` when E == H). */
int mh_cast_to_f32(const void* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows, int64_t cols,
                   int dtype, mh_stream_t stream);

/* Row-major fp32 [rows, cols] -> K32-panel bf16 ([cols_pad/32][ld_rows][32], zero fill) and back
 * (panel bf16 -> row-major fp32).  The panel layout is described at mh_gemm_bias_act_ex. */
int mh_pack_panel(const float* in, int64_t ld_in, void* out, int64_t ld_rows, int64_t rows, int64_t cols,
                  int64_t cols_pad, mh_stream_t stream);
int mh_unpack_panel_f32(const void* in, int64_t ld_rows, float* out, int64_t ld_out, int64_t rows, int64_t cols,
                        mh_stream_t stream);

/* out[r] = sum_c table[r,c]^2  (rounding.py:22: emb_norm) */
int mh_row_sqnorm(const float* table, float* out, int V, int E, mh_stream_t stream);

/* ------------------------------------------------------------------ denoiser building blocks */

/* K11  word_embedding(ids): out[n,:] = table[ids[n],:]   (models/network.py:88-89) */
int mh_embed_gather(const float* table, const int32_t* ids, float* out, int64_t n_tokens, int E, int V,
                    mh_stream_t stream);

/* K1   sinusoidal timestep embedding, fractional t allowed: out[b] = [cos(t f) | sin(t f) | 0 pad]
 *      f_i = exp(-ln(max_period) i / half)               (models/network.py:108-129)
 *      out is [B, ld_out] of `dtype`, columns >= dim are zero-filled up to ld_out. */
int mh_timestep_embedding(const float* t, void* out, int B, int dim, int64_t ld_out, float max_period,
                          int dtype, mh_stream_t stream);

/* K2/K3/K5/K7/K8/K9  out = act(A W^T + bias) [+ residual]   (nn.Linear sites: models/network.py:60-72,
 *      :81-86 and the HF BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput dense
 *      layers reached from network.py:151).
 *      A [M, lda] and W [N, ldw] are `dtype`; K must be a multiple of 64 (bf16) / 16 (f32) and the
 *      padded columns of A and W must be zero.  bias [N] fp32 or NULL.  residual [M, ldr] `dtype`
 *      or NULL (added after the activation).  out is `dtype`, or fp32 when out_f32 != 0. */
int mh_gemm_bias_act(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                     const void* residual, int64_t ldr, void* out, int64_t ldo, int out_f32,
                     int64_t M, int N, int K, int act, int dtype, mh_stream_t stream);

/* Same with explicit operand layouts (bf16 big-tile kernel only).  `*_panel != 0` selects the K32-PANEL
 * layout for that operand: element (r, c) of an [R, C] matrix lives at ((c / 32) * ld + r) * 32 + c % 32
 * with ld = allocated rows per panel (C padded to a multiple of 32).  In that layout every K-step of a
 * 256-row tile is one contiguous 16 KiB run, so each global_load_lds instruction moves 1 KiB of whole
 * cache lines (row-major needs sixteen 64-byte row segments per instruction) and epilogue stores of
 * adjacent rows are adjacent in memory. */
int mh_gemm_bias_act_ex(const void* A, int64_t lda, int a_panel, const void* W, int64_t ldw, int w_panel,
                        const float* bias, const void* residual, int64_t ldr, int r_panel, void* out, int64_t ldo,
                        int o_panel, int out_f32, int64_t M, int N, int K, int act, int dtype, mh_stream_t stream);

/* K5   fused Q/K/V projection: [q|k|v] = A Wqkv^T + bqkv with Wqkv = [Wq; Wk; Wv] ([3H, ldw]);
 *      scatters heads: q,k -> [B, nh, L, dh], v -> TRANSPOSED [B, nh, dh, L] (the layout the
 *      attention kernel's P.V product wants).  HF BertSelfAttention.{query,key,value} +
 *      transpose_for_scores.  L must be a multiple of 8. */
int mh_gemm_qkv(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, const float* bqkv, void* q,
                void* k, void* vt, int B, int L, int H, int nh, int dtype, mh_stream_t stream);

/* Batched out[b] = A[b] W[b]^T (+ bias): `batch` independent problems `stride*` elements apart (training path:
 * per-(batch, head) attention backward products).  Row-major operands. */
int mh_gemm_batched(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW,
                    const float* bias, void* out, int64_t ldo, int64_t strideO, int out_f32, int batch, int64_t M, int N,
                    int K, int dtype, mh_stream_t stream);

/* mh_gemm_qkv with A and/or Wqkv in the K32-panel layout (bf16 big-tile kernel). */
int mh_gemm_qkv_ex(const void* A, int64_t lda, int a_panel, const void* Wqkv, int64_t ldw, int w_panel,
                   const float* bqkv, void* q, void* k, void* vt, int B, int L, int H, int nh, int dtype,
                   mh_stream_t stream);

/* K6   ctx = softmax(q k^T * scale) v, no mask (none is ever passed: diffusion.py:309, network.py:151),
 *      flash-style (scores never materialised).  q,k [B,nh,L,dh], vt [B,nh,dh,L], ctx [B*L, ld_ctx]
 *      with head h at columns [h*dh, (h+1)*dh).  dh in {32,64,128} (bf16) / {16,32,64,128} (f32). */
int mh_attention_fwd(const void* q, const void* k, const void* vt, void* ctx, int64_t ld_ctx, int B, int L,
                     int nh, int dh, float scale, int dtype, mh_stream_t stream);

/* mh_attention_fwd writing ctx in the K32-panel layout when ctx_panel != 0 (ld_ctx = rows per panel). */
int mh_attention_fwd_ex(const void* q, const void* k, const void* vt, void* ctx, int64_t ld_ctx, int ctx_panel, int B,
                        int L, int nh, int dh, float scale, int dtype, mh_stream_t stream);

/* A/B switch: 1 (default) lets the bf16 kernel keep K and V^T of a (batch, head) resident in LDS when they
 * fit (<= 128 KiB), 0 forces the tiled double-buffered kernel. */
/* Streaming attention (bf16): same result as mh_attention_fwd_ex, for seq_len % 16 == 0, seq_len >= 512 (the reference's
 * default 2096 included: the last stage / key tile is masked), head dim 32 / 64 (mh_attention_stream_supported).  K / V^T are streamed through LDS by LDS-DMA in 256-key stages,
 * double-buffered across stages and (batch, head) items by persistent 16-wave blocks.  `vt_perm` is V^T
 * [B, nh, dh, L] with the keys of every group of 16 stored as 0-3, 8-11, 4-7, 12-15 (what mh_gemm_qkv_vtperm
 * writes): the order in which the S^T accumulators hold the probabilities.  Replaces BertSelfAttention of
 * the encoder called at models/network.py:151. */
int mh_attention_stream_fwd(const void* q, const void* k, const void* vt_perm, void* ctx, int64_t ld_ctx, int ctx_panel,
                            int B, int L, int nh, int dh, float scale, mh_stream_t stream);
/* mh_attention_stream_fwd that also writes lse2 [B, nh, L] fp32 = log2-domain log-sum-exp of the scaled scores
 * (P[q][k] = exp2(s[q][k] scale log2e - lse2[q])): what mh_attention_stream_bwd re-creates P from. */
int mh_attention_stream_fwd_lse(const void* q, const void* k, const void* vt_perm, void* ctx, int64_t ld_ctx, int ctx_panel,
                                int B, int L, int nh, int dh, float scale, float* lse2, mh_stream_t stream);
/* Fused attention backward (bf16; seq_len % 16 == 0, >= 512: partial last tiles are masked, as in the streaming forward).  q, k, v, dO: [B, nh, L, dh] rows;
 * qT_perm, kT_perm, dOT_perm: UNUSED since round 3 (may be null) - the kernels read K^T, Q^T and dO^T out of the row-layout LDS
 * stages with the transposing ds_read_b64_tr_b16, so the [B, nh, dh, L] copies of round 2 are no longer made; the parameters stay
 * for binary compatibility.  o = the forward output in dO's layout; lse2 from the forward; D = [B, nh, L] fp32 scratch (the dQ kernel
 * writes D[b, h, l] = sum_d dO o O there - times (1 - p) in the dropout variant - and the dK/dV kernel reads it).  dq, dk, dv are written
 * token-major: element (token, head, d) at ptr[token * ld_d + head * dh + d] - e.g. the three column blocks of a
 * [B L, 3H] gradient of the fused QKV projection.  Replaces autograd through BertSelfAttention (the encoder that
 * models/network.py:151 calls) inside training_losses (models/diffusion.py:594-699). */
int mh_attention_stream_bwd(const void* q, const void* k, const void* v, const void* qT_perm, const void* kT_perm,
                            const void* dO, const void* dOT_perm, const void* o, const float* lse2, float* D, void* dq, void* dk,
                            void* dv, int64_t ld_d, int B, int L, int nh, int dh, float scale, mh_stream_t stream);
/* The same two entry points for row operands that are NOT [B, nh, L, dh] tensors: row r of (batch b, head h) of
 * q / k (/ v) starts at ptr + b * batch_stride + h * head_stride + r * row_stride elements (all multiples of 8), e.g.
 * the column blocks of a token-major [B L, 3H] QKV projection: batch_stride L * 3H, head_stride dh, row_stride 3H. */
int mh_attention_stream_fwd_ex(const void* q, const void* k, const void* vt_perm, void* ctx, int64_t ld_ctx, int ctx_panel,
                               int B, int L, int nh, int dh, float scale, float* lse2, int64_t qk_batch_stride,
                               int64_t qk_head_stride, int64_t qk_row_stride, mh_stream_t stream);
int mh_attention_stream_bwd_ex(const void* q, const void* k, const void* v, const void* qT_perm, const void* kT_perm,
                               const void* dO, const void* dOT_perm, const void* o, const float* lse2, float* D, void* dq, void* dk,
                               void* dv, int64_t ld_d, int B, int L, int nh, int dh, float scale, int64_t qkv_batch_stride,
                               int64_t qkv_head_stride, int64_t qkv_row_stride, int64_t do_batch_stride,
                               int64_t do_head_stride, int64_t do_row_stride, mh_stream_t stream);
int mh_attention_stream_bwd_supported(int L, int dh);   /* seq_len % 16 == 0, >= 512, head dim 32 / 64 */
int mh_attention_bwd_rowdot(const void* dctx, const void* ctx, int64_t ld, float* D, int B, int L, int nh, int dh,
                            mh_stream_t stream);
int mh_attention_stream_supported(int L, int dh);
int mh_attention_stream_enabled(void);
/* mh_gemm_qkv_ex (bf16) writing V^T in the key order mh_attention_stream_fwd reads (seq_len % 16 == 0). */
int mh_gemm_qkv_vtperm(const void* A, int64_t lda, int a_panel, const void* Wqkv, int64_t ldw, int w_panel,
                       const float* bqkv, void* q, void* k, void* vt_perm, int B, int L, int H, int nh,
                       mh_stream_t stream);

/* K7/K8 tail  out = LayerNorm(x) * gamma + beta over the last dim (eps 1e-12 in the reference,
 *      network.py:79 and HF BertSelfOutput/BertOutput).  x, out [rows, H] `dtype`. */
int mh_layernorm(const void* x, const float* gamma, const float* beta, void* out, int64_t rows, int H,
                 float eps, int dtype, mh_stream_t stream);

/* K4   out[b,l,:] = LayerNorm(pos[l,:] + x[b,l,:] + emb_t[emb_row[b],:])   (models/network.py:146-149)
 *      x is `dtype` [B*L, ldx], or the fp32 latent itself when x_is_f32 (E == H: no up-projection).
 *      pos [L,H] fp32, emb_t [*,H] fp32, emb_row [B] int32 (NULL: row b). */
int mh_add_pos_time_layernorm(const void* x, int64_t ldx, int x_is_f32, const float* pos, const float* emb_t,
                              const int32_t* emb_row, const float* gamma, const float* beta, void* out,
                              int B, int L, int H, float eps, int dtype, mh_stream_t stream);

/* The two LayerNorm entry points on K32-panel bf16 activations (x may also be the row-major fp32 latent for
 * the add variant).  One wave normalises 16 rows; every access is a 1-KiB contiguous run. */
int mh_layernorm_panel(const void* x, int64_t ldx, const float* gamma, const float* beta, void* out, int64_t ldo,
                       int64_t rows, int H, float eps, mh_stream_t stream);
int mh_add_pos_time_layernorm_panel(const void* x, int64_t ldx, int x_is_f32, const float* pos, const float* emb_t,
                                    const int32_t* emb_row, const float* gamma, const float* beta, void* out,
                                    int64_t ldo, int B, int L, int H, float eps, mh_stream_t stream);

/* ------------------------------------------------------------------ rounding / logits (always fp32) */

/* K10  idx[n] = argmin_v clamp(|W_v|^2 + |x_n|^2 - 2 W_v.x_n, 0), first index on ties
 *      (models/rounding.py:21-28).  table [V,E], table_norm [V] from mh_row_sqnorm. */
int mh_round_to_embedding(const float* x, const float* table, const float* table_norm, int32_t* idx,
                          int64_t n_tokens, int E, int V, mh_stream_t stream);

/* K10 on the exact-fp32 MFMA (the per-step path): same result as mh_round_to_embedding, computed as a
 * 128x128-tiled v_mfma_f32_16x16x4_f32 GEMM whose epilogue keeps, per token, the best (score, first index)
 * of every 64-column slot, followed by a tiny cross-slot reduce.  table_pad is the table with rows
 * zero-padded to a multiple of 16 columns (the table itself when E % 16 == 0). */
size_t mh_round_workspace_bytes(int64_t n_tokens, int E, int V);
int mh_round_to_embedding_mfma(const float* x, const float* table_pad, const float* table_norm, int32_t* idx,
                               int64_t n_tokens, int E, int V, void* workspace, size_t workspace_bytes,
                               mh_stream_t stream);

/* K16  idx[n] = argmax_v (x_n . W_v + bias_v), first index on ties (run/sample.py:219-220 on
 *      models/network.py:91-93). */
int mh_logits_argmax(const float* x, const float* table, const float* bias, int32_t* idx, int64_t n_tokens,
                     int E, int V, mh_stream_t stream);

/* ------------------------------------------------------------------ diffusion arithmetic (fp32) */

/* K15  out = where(mask == 0, x0, a[b] * x0 + s[b] * noise)       (models/diffusion.py:229-255)
 *      a = sqrt_alphas_cumprod[t], s = sqrt_one_minus_alphas_cumprod[t] (host-extracted, [B] fp32).
 *      mask is int32 per TOKEN [B*L] (mask_per_elem == 0) or per ELEMENT [B*L*E], or NULL. */
int mh_q_sample(const float* x0, const float* noise, const float* a, const float* s, const int32_t* mask,
                int mask_per_elem, float* out, int B, int64_t per_batch, int E, mh_stream_t stream);

/* Per-step scalar coefficients of one reverse step, one struct per batch row or one shared row. */
typedef struct mh_step_coef {
  float coef1;      /* posterior_mean_coef1[t]                      (diffusion.py:265) */
  float coef2;      /* posterior_mean_coef2[t]                      (diffusion.py:266) */
  float sigma;      /* p_sample: (t != 0) * exp(0.5 * log_variance[t])   (diffusion.py:390-393)
                       ddim:     (t != 0) * sigma_t                      (diffusion.py:732-747) */
  float recip;      /* sqrt_recip_alphas_cumprod[t]                 (diffusion.py:203) */
  float recipm1;    /* sqrt_recipm1_alphas_cumprod[t]               (diffusion.py:205) */
  float sqrt_abp;   /* sqrt(alphas_cumprod_prev[t])                 (diffusion.py:740) */
  float dir;        /* sqrt(1 - alphas_cumprod_prev[t] - sigma_t^2) (diffusion.py:741) */
  float pad;
} mh_step_coef;

/* K10+K12  fused p_sample tail (models/diffusion.py:319-347, :390-397):
 *      x0   = round_idx ? table[round_idx[n]] : model_out       (denoised_fn, rounding.py:45)
 *      x0   = clip ? clamp(x0, -1, 1) : x0
 *      mean = coef1 * x0 + coef2 * x_t
 *      out  = mean + sigma * noise ;  out = where(mask == 0, x_start, out)
 *      coef is [1] (coef_per_batch == 0) or [B].  pred_xstart / mean_out may be NULL.
 *      out may alias x_t (in-place step). */
int mh_p_sample_epilogue(const float* model_out, const float* x_t, const float* noise, const int32_t* round_idx,
                         const float* table, const mh_step_coef* coef, int coef_per_batch, int clip,
                         const int32_t* mask, int mask_per_elem, const float* x_start, float* out,
                         float* pred_xstart, float* mean_out, int B, int64_t per_batch, int E,
                         mh_stream_t stream);

/* K10+K13  fused ddim_sample tail (models/diffusion.py:729-757):
 *      eps  = (recip * x_t - x0) / recipm1
 *      out  = x0 * sqrt_abp + dir * eps + sigma * noise ; anchoring as above. */
int mh_ddim_epilogue(const float* model_out, const float* x_t, const float* noise, const int32_t* round_idx,
                     const float* table, const mh_step_coef* coef, int coef_per_batch, int clip,
                     const int32_t* mask, int mask_per_elem, const float* x_start, float* out,
                     float* pred_xstart, int B, int64_t per_batch, int E, mh_stream_t stream);

/* K14  truncated standard normal, counter-based (Philox4x32-10): element n of call (seed, stream_id,
 *      *step_counter) is a pure function of those numbers, rejection (|z| > bound, bound <= 0: none)
 *      is resolved per element in registers: no host sync (models/diffusion.py:378-388).
 *      step_counter: device uint32 (may be NULL = 0); lets a captured graph advance the stream. */
int mh_trunc_normal(float* out, int64_t n, float bound, uint64_t seed, uint32_t stream_id,
                    const uint32_t* step_counter, mh_stream_t stream);
/*      the same for elements first .. first + n - 1 of the call (first % 4 == 0): a batch slice drawn on its own gets the values
 *      the whole-batch call gives it */
int mh_trunc_normal_at(float* out, int64_t n, int64_t first, float bound, uint64_t seed, uint32_t stream_id,
                       const uint32_t* step_counter, mh_stream_t stream);
/* one idle wave for about `microseconds` on `stream`: a phase lag between concurrently replayed graph branches */
int mh_stream_delay(unsigned microseconds, mh_stream_t stream);

/* ------------------------------------------------------------------ training path (forward + backward pieces)
 * training_losses (models/diffusion.py:594-699) under utils/train_util.py:188-232.  Every backward matrix
 * product is brought to the forward's "A W^T" form by the permutes / transposes below and runs on the same
 * GEMM kernels; reductions are fp32 and deterministic (two-stage) except the embedding scatter-add. */

/* out[b][c][r] = in[b][r][c] */
int mh_transpose(const void* in, int64_t ld_in, int64_t stride_in, void* out, int64_t ld_out, int64_t stride_out, int rows,
                 int cols, int batch, int dtype, mh_stream_t stream);
/* mode 0: tokens [B*L, ld_tok] -> heads [B,nh,L,dh]; 1: heads -> tokens; 2: tokens -> transposed heads [B,nh,dh,L];
 * 3: as 2 with the positions of every group of 16 stored as 0-3, 8-11, 4-7, 12-15 (streaming attention operand order);
 * 4: as 3, and the 256 elements behind the last row are zeroed (the slack the streaming kernels' 16-byte tail reads want: `out`
 * must hold B nh dh L + 256 elements) - bf16, L % 64 == 0, dh in {32, 64, 128} only */
int mh_head_permute(const void* in, void* out, int64_t ld_tok, int B, int L, int nh, int dh, int mode, int dtype,
                    mh_stream_t stream);
/* out[b, c] (+)= sum_r in[b][r, c]  (bias / LayerNorm-parameter / position / time-embedding gradients);
 * partial: [batch, n_partial, cols] fp32 scratch */
int mh_col_sum(const void* in, int64_t ld, int64_t rows, int cols, int batch, int64_t stride_in, float* partial, int n_partial,
               float* out, int accumulate, int dtype, mh_stream_t stream);
/* out[b,l,:] = pos[l,:] + x[b,l,:] + emb[b,:]  (network.py:146-148; the training path keeps this pre-LayerNorm sum) */
int mh_add_pos_time(const void* x, int64_t ldx, const float* pos, const float* emb, void* out, int B, int L, int H, int dtype,
                    mh_stream_t stream);
/* y = act(x) and dx = dy * act'(x) with x the saved PRE-activation (tanh / exact-erf GELU / SiLU) */
int mh_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, mh_stream_t stream);
int mh_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, int dtype, mh_stream_t stream);
/* LayerNorm backward: dx, and dgamma / dbeta (+)= ...; partial: [2, n_partial, H] fp32 scratch */
int mh_layernorm_bwd(const void* x, const void* dy, const float* gamma, void* dx, float* partial, int n_partial, float* dgamma,
                     float* dbeta, int accumulate, int64_t rows, int H, float eps, int dtype, mh_stream_t stream);
/* attention probabilities: P = softmax(S * scale) in place; dS = P o (dP - rowsum(dP o P)) * scale in place on dP */
int mh_softmax_rows(void* s_inout, int64_t rows, int L, int64_t ld, float scale, int dtype, mh_stream_t stream);
int mh_softmax_bwd_rows(const void* p, void* dp_inout, int64_t rows, int L, int64_t ld, float scale, int dtype, mh_stream_t stream);
/* token cross-entropy over fp32 logits [n, V] (diffusion.py:556-575): loss[n] = lse - logit[target];
 * dlogits = (softmax - onehot) * grad[n], written in `dtype` with columns [V, Vpad) zeroed */
int mh_cross_entropy_fwd(const float* logits, int64_t ld, const int32_t* target, float* loss, float* lse, int64_t n, int V,
                         mh_stream_t stream);
int mh_cross_entropy_bwd(const float* logits, int64_t ld, const int32_t* target, const float* lse, const float* grad, void* dlogits,
                         int64_t ldd, int64_t n, int V, int Vpad, int dtype, mh_stream_t stream);
/* mean_flat((scale_a * a - b)^2) per batch row (diffusion.py:627-639; b NULL = 0) and its gradient */
int mh_sqdiff_mean(const float* a, const float* b, float scale_a, float* out, int B, int64_t per_batch, mh_stream_t stream);
int mh_sqdiff_bwd(const float* a, const float* b, float scale_a, const float* grad, float* da, float* db, int accumulate, int B,
                  int64_t per_batch, mh_stream_t stream);
int mh_add_inplace(void* dst, const void* src, int64_t n, int dtype, mh_stream_t stream);
/* table[ids[n], :] += src[n, :]  (word_embedding gradient): chunked partial sums + ordered fold, no atomics,
 * reproducible; workspace of mh_scatter_add_rows_workspace_bytes(E, V) bytes */
int mh_scatter_add_rows(const float* src, const int32_t* ids, float* table, int64_t n, int E, int V, void* workspace,
                        size_t workspace_bytes, mh_stream_t stream);
size_t mh_scatter_add_rows_workspace_bytes(int E, int V);
/* q_sample backward: dst[b,i] (+)= src[b,i] * (mask[token] == 0 ? 1 : scale[b]) */
/* out[i] = sum_s in[s * n + i] (fp32, n % 4 == 0): folds the split-K partials of a weight-gradient GEMM
 * (mh_gemm_batched over K slices), summation order fixed -> reproducible gradients. */
int mh_sum_slices(const float* in, int slices, int64_t n, float* out, mh_stream_t stream);
int mh_scale_rows(const float* src, const float* scale, const int32_t* mask, float* dst, int accumulate, int B, int64_t per_batch,
                  int E, mh_stream_t stream);

/* bf16 working copies of fp32 master weights, all in ONE launch: for every item, dst [rows, ld_dst] = bf16(src [rows, cols]) and
 * (dst_t != NULL) dst_t [cols, ld_t] = its transpose - the operands of a dense layer's forward / weight-gradient GEMMs and of its
 * input-gradient GEMM.  Replaces the per-layer fp32 -> compute-dtype casts autocast-style training does on every forward
 * (the reference trains in fp32, utils/train_util.py:188-232; bf16 compute is this framework's mode).  rows, cols multiples of 64;
 * `items` is a DEVICE table, tile_start = running sum of (rows / 64) (cols / 64) over the preceding items, total_tiles its end. */
typedef struct mh_wprep_item {
  const float* src; void* dst; void* dst_t;
  int32_t rows, cols; int64_t ld_dst, ld_t; int32_t tile_start, pad_;
} mh_wprep_item;
int mh_weight_prep(const mh_wprep_item* items, int n_items, int total_tiles, mh_stream_t stream);

/* ------------------------------------------------------------------ optimizer step
 * Fused multi-tensor AdamW + up to 4 EMA copies (+ gradient L2 norm) over every parameter tensor in one
 * launch each: utils/train_util.py:246-280 (optimize, _log_grad_norm) and :21-31 (update_ema).
 * `tensors` / `chunks` are DEVICE tables built once by the host.  A tensor whose `grad` is NULL (a frozen parameter, e.g. the
 * reference's --freeze_embedding, config/train.py:67-68, or one autograd never reached) is skipped by the AdamW arithmetic exactly
 * as torch.optim.AdamW skips `p.grad is None` (train_util.py:95), adds nothing to the norm (:277) and is left alone by the clip,
 * while its EMA copies still take update_ema's step (:21-31 walks every master parameter).  param / grad / moments / EMA
 * pointers must be 16-byte aligned (the host checks). */
typedef struct mh_opt_tensor {
  float* param; const float* grad; float* exp_avg; float* exp_avg_sq; float* ema[4];
} mh_opt_tensor;
typedef struct mh_opt_chunk { int32_t tensor; int32_t count; int64_t offset; } mh_opt_chunk;
typedef struct mh_opt_hparams {   /* all derived scalars are evaluated in double by the host (as torch does) */
  float beta1, beta2, eps;
  float one_minus_beta1, one_minus_beta2;
  float decay_mul;   /* 1 - lr * weight_decay */
  float step_size;   /* lr / (1 - beta1^step) */
  float bias2_sqrt;  /* sqrt(1 - beta2^step) */
  int n_ema;
  float ema_rate[4];
  float ema_one_minus[4];
} mh_opt_hparams;
int mh_adamw_ema_step(const mh_opt_tensor* tensors, const mh_opt_chunk* chunks, int n_chunks, const mh_opt_hparams* hp,
                      mh_stream_t stream);
/* out[0] = sqrt(sum over all gradient elements of g^2); partial: [n_chunks] fp32 scratch */
int mh_grad_norm(const mh_opt_tensor* tensors, const mh_opt_chunk* chunks, int n_chunks, float* partial, float* out,
                 mh_stream_t stream);
/* torch.nn.utils.clip_grad_norm_ on the device (utils/train_util.py:255-264): every gradient *= min(1, max_norm / (norm[0] + 1e-6)),
 * `norm` = the device scalar mh_grad_norm wrote; no host synchronisation */
int mh_clip_grads(const mh_opt_tensor* tensors, const mh_opt_chunk* chunks, int n_chunks, const float* norm, float max_norm,
                  mh_stream_t stream);

/* ------------------------------------------------------------------ captured reverse step support */

/* Device-side loop state of a replayed reverse-diffusion step.  `steps` holds the timestep index
 * of every iteration in loop order (diffusion.py:508 / :878); pos advances once per replay. */
typedef struct mh_loop_state {
  uint32_t pos;        /* iteration counter (also the RNG step counter) */
  uint32_t n_steps;
  int32_t cur_t;       /* steps[pos], written by mh_step_begin */
  uint32_t rng_step;   /* mh_step_advance: the iteration number of the step in flight (pos already counts the next one) */
} mh_loop_state;

/* Reads state->pos, looks up t = steps[pos], writes state->cur_t, emb_row[0..B) = t and
 * *cur_coef = coef_table[t].  First node of a captured step. */
int mh_step_begin(mh_loop_state* state, const int32_t* steps, const mh_step_coef* coef_table,
                  mh_step_coef* cur_coef, int32_t* emb_row, int B, mh_stream_t stream);
/* state->pos += 1.  Last node of a captured step. */
int mh_step_end(mh_loop_state* state, mh_stream_t stream);
/* mh_step_begin and mh_step_end as ONE first node: looks up the step as mh_step_begin does, stores its iteration number in
 * state->rng_step (the counter the in-graph noise reads: pass &state->rng_step to mh_trunc_normal) and advances state->pos at once,
 * so that nothing follows the step's last kernel. */
int mh_step_advance(mh_loop_state* state, const int32_t* steps, const mh_step_coef* coef_table, mh_step_coef* cur_coef,
                    int32_t* emb_row, int B, mh_stream_t stream);

/* Thin hipGraph wrappers so the host can capture a sequence of the calls above on `stream` and
 * replay it (hipStreamBeginCapture / EndCapture / GraphInstantiate / GraphLaunch). */
/* out = act(A W^T + bias) and pre_out = A W^T + bias in one pass (bf16 row-major; shapes the big-tile kernel
 * serves: N % 8 == 0, K % 32 == 0, lda / ldw / ldo % 8 == 0; error otherwise).  Forward of dense + GELU / tanh under
 * autograd: the backward needs the pre-activation (training_losses, models/diffusion.py:594-699). */
int mh_gemm_bias_act_pre(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* pre_out,
                         void* out, int64_t ldo, int64_t M, int N, int K, int act, mh_stream_t stream);
/* mh_gemm_bias_act_pre for GELU with the DERIVATIVE stored in place of the pre-activation: dact_out = gelu'(A W^T + bias), out =
 * gelu(A W^T + bias), both bf16, the derivative from the same exp / rcp pair as the activation (round 3: the backward of
 * BertIntermediate -> BertOutput then multiplies by it - mh_gemm_act_grad with act = MH_ACT_DERIV - instead of evaluating gelu' from
 * the pre-activation: -22 ns of vector work per element in the backward for +6 in the forward). */
int mh_gemm_bias_act_dact(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* dact_out, void* out,
                          int64_t ldo, int64_t M, int N, int K, int act, mh_stream_t stream);
/* out = (A W^T) o act'(pre)  (act = MH_ACT_TANH or MH_ACT_GELU_ERF; bf16 row-major, big-tile shapes): the input-gradient
 * GEMM of the layer that FOLLOWS an activation, with the activation's backward in its epilogue - d(pre) of a dense + GELU
 * block without materialising d(activation) (training_losses backward, models/diffusion.py:594-699). */
int mh_gemm_act_grad(const void* A, int64_t lda, const void* W, int64_t ldw, const void* pre, int64_t ld_pre, void* out,
                     int64_t ldo, int64_t M, int N, int K, int act, mh_stream_t stream);
/* Weight gradient dW[M, N] = A^T B with both bf16 operands k-major as the forward left them (A [K, lda] = dY, B [K, ldb] = X,
 * K = tokens): no transposed copies; MFMA fragments come out of the k-major LDS image through ds_read_b64_tr_b16.  The token
 * range is cut into `splits` slices (mh_gemm_dw_splits) writing fp32 partials out_partials [splits][M][N]; fold them with
 * mh_sum_slices.  Replaces autograd's dense-layer weight gradients in training_losses (models/diffusion.py:594-699). */
int mh_gemm_dw(const void* A, int64_t lda, const void* B, int64_t ldb, float* out_partials, int splits, int64_t K, int M, int N,
               mh_stream_t stream);
int mh_gemm_dw_splits(int64_t K, int M, int N);
/* The same with the bias gradient of that linear as a by-product (with_colsum != 0): each split slice is M N + M floats - the
 * products, then sum_k A[k][m] over the slice's tokens, taken off the matrix pipe (an all-ones operand) by the blocks that already
 * hold the A panel - so dY is not read a second time for autograd's `grad_bias = dY.sum(0)`; one mh_sum_slices over M N + M folds both. */
int mh_gemm_dw_bias(const void* A, int64_t lda, const void* B, int64_t ldb, float* out_partials, int splits, int64_t K, int M, int N,
                    int with_colsum, mh_stream_t stream);
/* out = LayerNorm(A W^T + bias + residual) * gamma + beta, bf16, the whole row normalised inside the
 * GEMM epilogue (one block owns all N columns: N must be 128, 256 or 512 - see ..._supported).
 * Replaces BertSelfOutput / BertOutput (dense -> LayerNorm(hidden + input)) of the encoder that
 * models/network.py:150 calls.  Operands row-major or K32-panel as in mh_gemm_bias_act_ex. */
int mh_gemm_bias_res_ln(const void* A, int64_t lda, int a_panel, const void* W, int64_t ldw, int w_panel,
                        const float* bias, const void* residual, int64_t ldr, int r_panel, const float* gamma,
                        const float* beta, float eps, void* out, int64_t ldo, int o_panel, int64_t M, int N,
                        int K, mh_stream_t stream);
int mh_gemm_bias_res_ln_supported(int N);
int mh_denoiser_get_fuse_ln(void);

int mh_graph_begin_capture(mh_stream_t stream);
int mh_graph_end_capture(mh_stream_t stream, void** graph_exec_out);
int mh_graph_launch(void* graph_exec, mh_stream_t stream);
int mh_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------ whole denoiser forward */

typedef struct mh_layer_weights {
  const void* w_qkv;   const float* b_qkv;     /* [3H, H]  query/key/value stacked */
  const void* w_ao;    const float* b_ao;      /* attention.output.dense [H, H] */
  const float* ln1_g;  const float* ln1_b;     /* attention.output.LayerNorm */
  const void* w_ff1;   const float* b_ff1;     /* intermediate.dense [F, H] */
  const void* w_ff2;   const float* b_ff2;     /* output.dense [H, F] */
  const float* ln2_g;  const float* ln2_b;     /* output.LayerNorm */
  /* deferred LayerNorm (NULL: not packed): weights folded with the gain of the LayerNorm that feeds them, and their c1 / c2 vectors */
  const void* w_qkv_f; const float* c1_qkv; const float* c2_qkv;   /* gamma = PREVIOUS layer's output.LayerNorm (layer 0: unused) */
  const void* w_ff1_f; const float* c1_ff1; const float* c2_ff1;   /* gamma = this layer's attention.output.LayerNorm */
} mh_layer_weights;

typedef struct mh_denoiser {
  int dtype;           /* mh_dtype of every `const void*` weight and of the activations (MH_BF16X3 / MH_F16X3: split panels, see mh_split_*) */
  int E, H, F, nh, nL, Tt, Tt_pad, T4_pad, E_pad, L_max;  /* *_pad: padded to a multiple of 64 */
  int panel;           /* bf16 only: weights and activations in the K32-panel layout (see mh_gemm_bias_act_ex) */
  int has_proj;        /* E != H: input_up_proj / output_down_proj present (network.py:67-72, :81-86) */
  float ln_eps;
  const void* w_t0;    const float* b_t0;      /* time_embed.0 [4Tt, Tt_pad] */
  const void* w_t2;    const float* b_t2;      /* time_embed.2 [H, T4_pad] */
  const void* w_up0;   const float* b_up0;     /* input_up_proj.0 [H, E_pad] */
  const void* w_up2;   const float* b_up2;     /* input_up_proj.2 [H, H] */
  const float* pos;                            /* position_embeddings [L_max, H] fp32 */
  const float* ln0_g;  const float* ln0_b;     /* LayerNorm */
  const void* w_dn0;   const float* b_dn0;     /* output_down_proj.0 [H, H] */
  const void* w_dn2;   const float* b_dn2;     /* output_down_proj.2 [E, H] */
  const mh_layer_weights* layers;              /* HOST array of nL entries */
} mh_denoiser;

/* bytes of scratch mh_denoiser_forward needs for a [B, L] batch */
size_t mh_denoiser_workspace_bytes(const mh_denoiser* m, int B, int L);

/* emb_t[b,:] = time_embed(timestep_embedding(t[b]))   (models/network.py:139).  out [B,H] fp32. */
int mh_time_embed(const mh_denoiser* m, const float* t, float* emb_t_out, int B, void* workspace,
                  size_t workspace_bytes, mh_stream_t stream);

/* TransformerNetModel.forward (models/network.py:131-158) given emb_t rows:
 *   x [B,L,E] fp32 -> out [B,L,E] fp32.  emb_t [*,H] fp32, emb_row [B] int32 or NULL (row b). */
int mh_denoiser_forward(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row,
                        float* out, int B, int L, void* workspace, size_t workspace_bytes,
                        mh_stream_t stream);

/* ---- split precision: the mode between MH_BF16 (fast, ~4e-3 per product) and MH_F32 (exact, f32 matrix rate = 1/16) --------------
 * dtype MH_BF16X3 or MH_F16X3.  A "split panel" matrix [rows, C] is [2][C / 32][ld rows][32] 16-bit values: the K32-panel layout once
 * for hi = rn16(v) and once for lo = rn16(v - hi).  Products run as lo x hi + hi x lo + hi x hi on the bf16 / f16 matrix pipe with fp32
 * accumulation (relative error per product 2^-16 / 2^-22 against 2^-9 for bf16); everything between the GEMMs is fp32 (exact-erf GELU,
 * tanhf, two-pass LayerNorm) as in the MH_F32 mode.  Reference arithmetic: fp32 everywhere (models/network.py:131-158, diffusion.py:914).
 * mh_denoiser_forward / mh_time_embed accept a descriptor with one of these dtypes (weights: split panels [2][K_pad/32][rows][32] from
 * mh_split_pack; the time-MLP weights stay fp32 row-major; panel = 0; H % 64 == 0, head dim in {16, 32, 64}). */
int mh_split_supported(int dtype);
/* fp32 row-major [rows, cols] (ldx) -> split panels with the K dimension zero-padded to kpad (a multiple of 32) */
int mh_split_pack(const float* x, int64_t ldx, void* out, int64_t ld_rows, int64_t rows, int cols, int kpad, int dtype, mh_stream_t stream);
/* split panels (cpad / 32 panels per part) -> fp32 row-major [rows, cols]: hi + lo */
int mh_split_join(const void* in, int64_t ld_rows, float* out, int64_t ldo, int64_t rows, int cols, int cpad, int dtype, mh_stream_t stream);
/* LayerNorm of fp32 rows [rows, H] (ldx) -> split panels; pos != NULL: of (pos[row % L] + x) + emb_t[emb_row ? emb_row[row / L] : row / L]
 * first (network.py:148-149) */
int mh_split_layernorm(const float* x, int64_t ldx, const float* pos, const float* emb_t, const int32_t* emb_row, const float* gamma,
                       const float* beta, void* out, int64_t ld_rows, int64_t rows, int L, int H, float eps, int dtype, mh_stream_t stream);
/* out = act(A W^T + bias) [+ residual].  A: split panels, M rows, K columns (lda rows per panel); W: split panels, N rows (ldw);
 * bias fp32 [N] (bias_rows = 0) or [M] (bias_rows = 1: the transposed V projection W_v X^T); residual: split panels [M, N] (ldr) or NULL;
 * out_mode 0: split panels [2][N/32][ldo][32]; 1: split row-major [M][ldo], lo part out_part elements after hi; 2: fp32 row-major [M][ldo].
 * act: MH_ACT_NONE / MH_ACT_TANH (tanhf) / MH_ACT_GELU_ERF (erff). */
int mh_split_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int bias_rows, const void* residual, int64_t ldr,
                  void* out, int64_t ldo, int out_mode, int64_t out_part, int64_t M, int N, int K, int act, int dtype, mh_stream_t stream);
/* out = LayerNorm(A W^T + bias + residual) as split panels, in ONE kernel whose blocks own complete rows (HF BertSelfOutput / BertOutput:
 * dense -> + input -> LayerNorm; replaces mh_split_gemm (fp32 rows out) + mh_split_layernorm).  N must satisfy ..._supported (512). */
int mh_split_gemm_res_ln_supported(int N);
int mh_split_gemm_res_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual, int64_t ldr,
                         const float* gamma, const float* beta, float eps, void* out, int64_t ldo, int64_t M, int N, int K, int dtype,
                         mh_stream_t stream);
/* unmasked multi-head self-attention on split operands (HF BertSelfAttention from network.py:151): q / k split row-major
 * [2][B L][ld_qk] (q of head h at column h dh, k at column k_offset + h dh, lo part qk_part elements after hi), vt split row-major
 * [2][nh dh][ld_vt] (column = token; lo part vt_part elements after hi), ctx -> split panels [2][nh dh / 32][ld_ctx][32] */
int mh_split_attention(const void* qk, int64_t ld_qk, int k_offset, int64_t qk_part, const void* vt, int64_t ld_vt, int64_t vt_part, void* ctx,
                       int64_t ld_ctx, int B, int L, int nh, int dh, float scale, int dtype, mh_stream_t stream);

/* ---- batch producers and token validators.  Ragged int32 sequences:
 * `values` = all rows back to back, `offsets[B + 1]` int64; rows of at most mh_batch_max_row() tokens.
 * The reference draws its randomness from a host random.Random in data-dependent order; here every draw is an input,
 * indexed the way the reference consumes it (oracle/batch.py), so results are comparable bit for bit. */
int mh_batch_max_row(void);
/* data/wrapper.py:90-127 collate_batches, one field: out[b, j] = j < len_b ? values[offsets[b] + j] : pad; length[b] (may be NULL) */
int mh_ragged_to_padded(const int32_t* values, const int64_t* offsets, int32_t* out, int32_t* length, int B, int L,
                        int32_t pad, mh_stream_t stream);
/* utils/decode_util.py:221-230 meta_to_batch: ids[:, :len] = meta (else 0); mask = 0 on [:, :len + 1], 1 elsewhere */
int mh_meta_to_batch(const int32_t* meta, int len_meta, int32_t* ids, int32_t* mask, int B, int L, mh_stream_t stream);
/* data/corruption.py:100-114 masking_token: positions 12 .. first EOS - 1 become 0 where u[offsets[b] + (j - 12)] < p */
int mh_corrupt_masking_token(const int32_t* values, const int64_t* offsets, const float* u, float p, int32_t* out, int B,
                             mh_stream_t stream);
/* :117-133 masking_note: the k-th velocity token (131..194, idx + 3 <= len) zeroes [idx-1, idx+3) where u[offsets[b] + k] < p */
int mh_corrupt_masking_note(const int32_t* values, const int64_t* offsets, const float* u, float p, int32_t* out, int B,
                            mh_stream_t stream);
/* :136-162 randomize_note: ... replaces (velocity, pitch, duration) by new_tokens[(offsets[b] + k) * 3 + 0..2] */
int mh_corrupt_randomize_note(const int32_t* values, const int64_t* offsets, const float* u, const int32_t* new_tokens,
                              float p, int32_t* out, int B, mh_stream_t stream);
/* :165-195 random_rotating: `count` swaps of bars pairs[b][s] = (first < second); bar starts / last EOS taken once from the
 * input row as the reference does.  status[b] (may be NULL): 0 ok, 1 = fewer than two bars / no EOS / bad pair (row copied
 * up to the failing swap), 2 = row too long. */
int mh_corrupt_random_rotating(const int32_t* values, const int64_t* offsets, const int32_t* pairs, int count, int32_t* out,
                               int32_t* status, int B, mh_stream_t stream);
/* utils/decode_util.py:73-84, :142-183: per row of tokens [B, L] (valid length lens[b], NULL = L): result[b] = (index of the first
 * EOS or -1, validate_once passes, validate_rigidly passes | -2 where the reference raises IndexError on a truncated note) */
int mh_validate_tokens(const int32_t* tokens, const int32_t* lens, int32_t* result, int B, int L, mh_stream_t stream);

/* metric.py:4-71 get_vectors over a batch of note sequences tokens [B, L] (valid length lens[b], NULL = L):
 * out[b] = [32 rhythm | 12 melody | 12 harmony] fp32, each L2-normalised - the MSIM / 1NNC features.  status[b] (may be
 * NULL): 0 ok, 1 = no BAR token, 2 = malformed note group / no terminator (the reference raises ValueError / IndexError). */
int mh_msim_vectors(const int32_t* tokens, const int32_t* lens, float* out, int32_t* status, int B, int L, float note_len,
                    mh_stream_t stream);

/* metric.py:120-168 Controllability_Pitch / _Velocity, the per-row part: out[b] = (sum of pitch tokens 3..130, their count,
 * count of velocity tokens 131..194, count of those outside [metas[b][7] - 524, metas[b][8] - 524] with 130 / 195 = open bound);
 * tokens [B, L] note sequences (valid length lens[b], NULL = L), metas [B, meta_ld] meta tokens (meta_ld >= 9). */
int mh_controllability_counts(const int32_t* tokens, const int32_t* lens, const int32_t* metas, int meta_ld, int32_t* out, int B,
                              int L, mh_stream_t stream);

/* ---------------------------------------------------------------- train-mode dropout
 * Reference: nn.Dropout(dropout) after the embedding LayerNorm (MuseDiffusion/models/network.py:76, :149) and the HF BertEncoder's
 * hidden dropout (after the attention-output and FFN-output dense layers, before residual + LayerNorm) and attention-probability
 * dropout, both 0.1 from the bert-base config (network.py:44-46, :74).  Masks are counter-based (Philox4x32-7 keyed by `seed`,
 * counter = (element group, offset)): the backward re-creates a dense site's mask from the same descriptor; the attention mask
 * is written once by the forward as a bit tensor (1 bit per probability).  `mask` (test-only, dense sites): explicit keep flags,
 * one byte per element in row-major order, overriding Philox - how parity tests inject the oracle's masks.  p == 0 (or a null
 * descriptor): identity. */
typedef struct mh_dropout {
  float p;             /* drop probability in [0, 1) */
  uint64_t seed;       /* Philox key */
  uint64_t offset;     /* counter words 2..3: one value per (forward call, site) */
  const uint8_t* mask; /* optional explicit keep flags (device pointer) */
} mh_dropout;

/* out = x o keep / (1 - p) over [rows, cols] (cols % 8 == 0); element index = row * cols + col.  Forward of the embedding site,
 * backward of every dense site. */
int mh_dropout_fwd(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int cols, int dtype, const mh_dropout* drop,
                   mh_stream_t stream);
/* pre_out = dropout(A W^T + bias) + residual (bf16-rounded) AND out = LayerNorm(pre_out) * gamma + beta in ONE kernel: HF
 * BertSelfOutput / BertOutput in train mode (dense -> dropout -> LayerNorm(hidden + input)) on the full-row tile (N = 512,
 * row-major bf16).  pre_out is what the LayerNorm backward reads; the statistics are taken from the rounded values, i.e. what a
 * separate LayerNorm over pre_out computes.  drop may be NULL / p = 0 (eval, or a dropout-free fine-tune). */
int mh_gemm_bias_dropout_res_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual,
                                int64_t ldr, const float* gamma, const float* beta, float eps, void* pre_out, void* out, int64_t ldo,
                                int64_t M, int N, int K, const mh_dropout* drop, mh_stream_t stream);
/* out = dropout(A W^T + bias) + residual: BertSelfOutput / BertOutput dense -> dropout -> (+ input) in one GEMM (row-major). */
int mh_gemm_bias_dropout_res(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual,
                             int64_t ldr, void* out, int64_t ldo, int64_t M, int N, int K, int dtype, const mh_dropout* drop,
                             mh_stream_t stream);
/* attention-probability keep bits, 1 bit per probability, lane-native layout (csrc/common.h drop_word_index): uint32
 * [B nh][nb = ceil(L/32) query blocks][ceil(nb/2) pairs of 32-key blocks][64 lanes]; the word of lane (query % 32) + 32 h holds bit
 * r (+ 16 for the odd key block) <-> key (r & 3) + 8 (r >> 2) + 4 h of the block */
size_t mh_dropout_bits_words(int BH, int L);
int mh_dropout_bits(uint32_t* keep_bits, int BH, int L, const mh_dropout* drop, mh_stream_t stream);
/* mh_layernorm_bwd that also writes dx_dropped = dx o keep / (1 - p) for the dropout site of the dense layer IN FRONT of the LayerNorm
 * (BertSelfOutput / BertOutput: dense -> dropout -> + input -> LayerNorm): the dense branch of that node's backward reads dx_dropped,
 * the residual branch dx.  Element index of the mask = row * H + col (the dense sites' convention); identical to mh_dropout_fwd
 * applied to the stored dx.  drop == NULL or p == 0: mh_layernorm_bwd. */
int mh_layernorm_bwd_drop(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, const mh_dropout* drop,
                          float* partial, int n_partial, float* dgamma, float* dbeta, int accumulate, int64_t rows, int H, float eps,
                          int dtype, mh_stream_t stream);
/* P[bh][q][k] = keep ? P / (1 - p) : 0 over a materialised [B nh, L, ldp] tensor (probabilities forward, their gradient backward) */
int mh_dropout_bits_apply(void* P, int64_t ldp, const uint32_t* keep_bits, int BH, int L, float p, int dtype, mh_stream_t stream);
/* streaming attention with probability dropout: keep_bits written (bits_in = 0) or read (bits_in = 1) by the forward, read by the
 * backward (BertSelfAttention: softmax -> dropout -> . V) */
int mh_attention_stream_fwd_drop(const void* q, const void* k, const void* vt_perm, void* ctx, int64_t ld_ctx, int ctx_panel, int B,
                                 int L, int nh, int dh, float scale, float* lse2, int64_t qk_batch_stride, int64_t qk_head_stride,
                                 int64_t qk_row_stride, const mh_dropout* drop, uint32_t* keep_bits, int bits_in, mh_stream_t stream);
int mh_attention_stream_bwd_drop(const void* q, const void* k, const void* v, const void* qT_perm, const void* kT_perm, const void* dO,
                                 const void* dOT_perm, const void* o, const float* lse2, float* D, void* dq, void* dk, void* dv,
                                 int64_t ld_d, int B, int L, int nh, int dh, float scale, int64_t qkv_batch_stride,
                                 int64_t qkv_head_stride, int64_t qkv_row_stride, int64_t do_batch_stride, int64_t do_head_stride,
                                 int64_t do_row_stride, const uint32_t* keep_bits, float drop_p, mh_stream_t stream);

/* ---------------------------------------------------------------- deferred LayerNorm (bf16 throughput path)
 * HF BertSelfOutput / BertOutput end in LayerNorm(dense(x) + input).  Instead of normalising in the producing GEMM (which needs
 * a tile that owns complete rows) the producer stores the RAW sum and one partial (sum, sum of squares) pair per row and
 * 128-column tile (`o_stats`, [M][o_slots][2] fp32); consumers normalise on the fly:
 *   A operand raw (`a_stats`):  LN(y) W^T + b = rstd_r ((y W'^T)_rc - mean_r c1_c) + c2_c, W' = gamma o W (rows of the folded
 *       weight matrix passed as W), c1_c = sum_k W'_ck, and the `bias` argument = c2_c = sum_k beta_k W_ck + b_c;
 *   residual raw (`r_stats`):   (y - mean_r) rstd_r gamma_c + beta_c.
 * All operands in the K32-panel layout, 256x128 tile.  Reference: the LayerNorm calls inside network.py:151's BertEncoder. */
typedef struct mh_ln_defer {
  const float* a_stats; int a_slots; const float* c1;
  const float* r_stats; int r_slots; const float* r_gamma; const float* r_beta;
  float* o_stats; int o_slots;
  int h_norm;          /* width of the normalised rows (d_model) */
  float eps;
} mh_ln_defer;
int mh_gemm_bias_act_defer(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual, int64_t ldr,
                           void* out, int64_t ldo, int64_t M, int N, int K, int act, const mh_ln_defer* defer, mh_stream_t stream);
int mh_gemm_qkv_vtperm_defer(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, const float* c2, void* q, void* k, void* vt_perm,
                             int B, int L, int H, int nh, const mh_ln_defer* defer, mh_stream_t stream);
/* The forward in three phases over K32-panel activation buffers the caller owns (bf16 [H / 32][ld rows][32]; bf16 panel models with
 * up / down projections: mh_denoiser_phases_supported): head = latent -> up-projection -> + position / time -> LayerNorm
 * (network.py:141-149), layers = the encoder (network.py:151), tail = down-projection (network.py:153-157).  A batch slice passes a row
 * window of a full-batch buffer (pointer + first_row * 32 elements, ld = the full batch's rows), so head and tail can run once for the
 * whole batch while the encoder layers run per slice on concurrent streams.  Workspace: mh_denoiser_workspace_bytes(m, B, L) of the call. */
/* Head and tail of the denoiser as one kernel each (bf16 K32-panel operands; d_model 256 / 512; E_pad <= 128), a block owning 64 complete
 * rows with the [64, d_model] intermediate of the first dense layer kept in LDS (csrc/headtail.hip):
 *   mh_up_proj_ln_fused:  out = LayerNorm((pos[l] + (tanh(x W0^T + b0) W2^T + b2)) + emb_t[emb_row[b]]) * gamma + beta  - input_up_proj,
 *     position / time add and the embedding LayerNorm of models/network.py:141-149 (x [B L, E] fp32 row-major, W0 [E_pad / 32][H][32],
 *     W2 [H / 32][H][32], out bf16 [H / 32][ldo][32]); replaces mh_pack_panel + 2 x mh_gemm_bias_act_ex + mh_add_pos_time_layernorm_panel.
 *   mh_down_proj_fused:   out = tanh(X W0^T + b0) W2^T + b2  - output_down_proj of models/network.py:153-157 (X bf16 [H / 32][ldx][32],
 *     W2 [H / 32][E][32], out [rows, E] fp32 row-major); replaces 2 x mh_gemm_bias_act_ex.
 * ..._supported: 1 when the shape is served (the engine falls back to the separate launches otherwise). */
int mh_up_proj_ln_fused_supported(int E, int E_pad, int H);
int mh_down_proj_fused_supported(int E, int H);
int mh_up_proj_ln_fused(const float* x, int E, int E_pad, const void* w0, const float* b0, const void* w2, const float* b2,
                        const float* pos, const float* emb_t, const int32_t* emb_row, const float* gamma, const float* beta,
                        float eps, void* out, int64_t ldo, int B, int L, int H, mh_stream_t stream);
int mh_down_proj_fused(const void* X, int64_t ldx, const void* w0, const float* b0, const void* w2, const float* b2, float* out,
                       float* out_sqnorm /* optional [rows]: |out row|^2 */, int64_t rows, int E, int H, mh_stream_t stream);
/* mh_denoiser_forward that also writes |out row|^2 per token (the |x_n|^2 of the rounding scores, models/rounding.py:23) - available when
 * the forward ends in mh_down_proj_fused (mh_denoiser_gives_sqnorm) */
int mh_denoiser_gives_sqnorm(const mh_denoiser* m);
int mh_denoiser_forward_sqnorm(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row, float* out,
                               float* out_sqnorm, int B, int L, void* workspace, size_t workspace_bytes, mh_stream_t stream);
/* Rounding INSIDE the forward's last kernel (bf16 throughput mode; E = 128, d_model 512, 640 < V <= 768 - the ComMU vocabulary on
 * BASELINE config 2's width): the fused down-projection splits its fp32 rows into bf16 hi + lo parts and contracts
 * [x_hi | x_lo | x_hi] with the table's [T_hi | T_hi | T_lo] on the bf16 matrix pipe (three products per element, fp32 accumulation:
 * fp32-grade scores to 2^-16), then takes -(clamp((|T_v|^2 + |x_n|^2) - 2 x.T_v, 0)) and the first-index argmax of
 * models/rounding.py:21-28.  mh_round_split_table prepares the table operand once per loop (buf: mh_round_split_bytes; table_norm:
 * optional caller-computed |T_v|^2 [V]).  The fp32 parity mode keeps the exact-fp32 score GEMM (mh_round_to_embedding_mfma). */
/* rng != NULL: the update kernel draws the step's noise itself - for element group (first_elem / 4 + g) exactly the four values
 * mh_trunc_normal_at(out, n, first_elem, bound, seed, stream_id, step_counter) writes (diffusion.py:378-388 top-p rejection resolved in
 * registers; bound 0 = plain normals) - and `noise` is not read. */
typedef struct mh_step_rng {
  uint64_t seed; uint32_t stream_id; float bound; const uint32_t* step_counter; int64_t first_elem;
} mh_step_rng;
/* upd != NULL (mh_down_proj_round_fused, mh_denoiser_forward_round): the same kernel also takes the reverse step of its rows - the
 * arithmetic of mh_p_sample_epilogue (ddim = 0) / mh_ddim_epilogue (ddim = 1) with x0 = table[nearest row]: x is x_t on entry and
 * x_{t-1} on return (in place), rng as in mh_step_epilogue_slots (NULL: `noise` is read, or no noise when that is NULL too). */
typedef struct mh_step_update {
  float* x; const float* x_start; const int32_t* mask; int mask_per_elem;
  const float* table; const mh_step_coef* coef; int clip; int ddim;
  float* pred_xstart; float* mean_out; const float* noise; const mh_step_rng* rng;
} mh_step_update;
int mh_down_proj_round_supported(int E, int H, int V);
size_t mh_round_split_bytes(int E, int V);
int mh_round_split_table(const float* table, const float* table_norm, int V, int E, void* buf, mh_stream_t stream);
int mh_down_proj_round_fused(const void* X, int64_t ldx, const void* w0, const float* b0, const void* w2, const float* b2, float* out,
                             float* out_sqnorm, const void* table_split, int V, int32_t* idx_out, const mh_step_update* upd, int64_t rows,
                             int E, int H, mh_stream_t stream);
int mh_denoiser_rounds_in_forward(const mh_denoiser* m, int V);
int mh_denoiser_forward_round(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row, float* out,
                              const void* table_split, int V, int32_t* idx_out, const mh_step_update* upd, int B, int L, void* workspace,
                              size_t workspace_bytes, mh_stream_t stream);
/* Rounding (models/rounding.py:21-28) split for a captured step: mh_round_scores = the exact-fp32 score GEMM of
 * mh_round_to_embedding_mfma with |x_n|^2 supplied by the caller, leaving the winners of every column slot in pbest / pidx
 * [n_tokens][mh_round_slots(V)]; mh_step_epilogue_slots = mh_p_sample_epilogue (ddim = 0) / mh_ddim_epilogue (ddim = 1) folding
 * those slots itself (larger score, then smaller index - the rule of the separate reduce) and optionally writing the index per row;
 * nslots = 0: pidx already holds the final index per row (mh_denoiser_forward_round) and pbest is not read.
 * Together with mh_denoiser_forward_sqnorm: two launches less per batch slice and step.  E % 16 == 0. */
int mh_round_slots(int V);
int mh_round_scores(const float* x, const float* x_sqnorm, const float* table_pad, const float* table_norm, float* pbest,
                    int32_t* pidx, int64_t n_tokens, int E, int V, mh_stream_t stream);
int mh_step_epilogue_slots(int ddim, const float* x_t, const float* noise, const float* pbest, const int32_t* pidx, int nslots,
                           const float* table, const mh_step_coef* coef, int coef_per_batch, int clip, const int32_t* mask,
                           int mask_per_elem, const float* x_start, float* out, float* pred_xstart, float* mean_out,
                           int32_t* round_idx_out, const mh_step_rng* rng, int B, int64_t per_batch, int E, mh_stream_t stream);
int mh_denoiser_phases_supported(const mh_denoiser* m);
int mh_denoiser_head(const mh_denoiser* m, const float* x, const float* emb_t, const int32_t* emb_row, void* x_out, int64_t ld_out, int B, int L,
                     void* workspace, size_t workspace_bytes, mh_stream_t stream);
int mh_denoiser_layers(const mh_denoiser* m, const void* x_in, int64_t ld_in, void* x_out, int64_t ld_out, int B, int L, void* workspace,
                       size_t workspace_bytes, mh_stream_t stream);
int mh_denoiser_tail(const mh_denoiser* m, const void* x_in, int64_t ld_in, float* out, int B, int L, void* workspace, size_t workspace_bytes,
                     mh_stream_t stream);
/* mh_gemm_qkv_vtperm over K32-panel operands with the queries stored as (x Wq^T + bq) * q_scale (rounded once, from the fp32
 * accumulator); `defer` (may be NULL) as in mh_gemm_qkv_vtperm_defer.  HF BertSelfAttention's projections (network.py:151) */
int mh_gemm_qkv_vtperm_qs(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, const float* bqkv, void* q, void* k, void* vt_perm,
                          int B, int L, int H, int nh, float q_scale, const mh_ln_defer* defer, mh_stream_t stream);
/* mh_attention_stream_fwd for queries that already carry softmax scale x log2(e) (q_scale above): the running softmax reference lives in
 * the initial value of the score accumulators, so a probability is exp2(accumulator) - no multiply-subtract per score */
int mh_attention_stream_prescaled_supported(int L, int dh);   /* mh_attention_stream_supported and seq_len % 256 == 0 */
int mh_attention_stream_fwd_prescaled(const void* q, const void* k, const void* vt_perm, void* ctx, int64_t ld_ctx, int ctx_panel,
                                      int B, int L, int nh, int dh, mh_stream_t stream);
int mh_denoiser_get_defer_ln(void);

/* get_logits with logits_mode 2 (models/network.py:94-104, "standard cosine similarity": in fact the negative Euclidean distance to every
 * lm_head row): out[n][v] = -sqrt(clamp((w_sqnorm[v] + x_sqnorm[n]) - 2 dots[n][v], 0, inf)); dots = x W^T from mh_gemm_bias_act (fp32). */
int mh_distance_scores(const float* dots, int64_t ld, const float* w_sqnorm, const float* x_sqnorm, float* out, int64_t ldo, int64_t n, int V,
                       mh_stream_t stream);

/* ---------------------------------------------------------------- training step on K32 panels (round 6)
 * The encoder layers of training_losses (models/diffusion.py:594-699 through models/network.py:151 -> HF BertLayer forward, and its
 * autograd backward) with every GEMM operand in the K32-panel layout of the sampler's kernels: the reference runs ~40 ATen kernels
 * per layer forward and as many backward; here one call per layer and direction launches the layer's kernels (csrc/train_layer.hip).
 *
 * mh_gemm_desc_launch: one descriptor for every bf16 dense launch of that path - out = [LayerNorm](drop(act(A W^T + bias)) [+ | o act'] residual):
 * each of A / W / residual / out / pre_out row-major (ld = row pitch) or K32 panels (ld = rows of the panel buffer); pre_out: the
 * pre-activation (pre_kind 0) or act'(pre) (1) of a dense + activation, in out's layout - or, with ln_gamma, the bf16-rounded
 * pre-LayerNorm rows in its own layout (ldp / p_panel) while out receives the normalised rows (N = 512, the full-row tile);
 * act_grad: `residual` holds a stored pre-activation / derivative and multiplies the product (mh_gemm_act_grad). */
typedef struct mh_gemm_desc {
  const void* A; int64_t lda; int a_panel;
  const void* W; int64_t ldw; int w_panel;
  const float* bias;
  const void* residual; int64_t ldr; int r_panel;
  void* out; int64_t ldo; int o_panel; int out_f32;
  void* pre_out; int64_t ldp; int p_panel; int pre_kind;
  int act, act_grad;
  const float* ln_gamma; const float* ln_beta; float ln_eps;
  const mh_dropout* drop;
  int64_t M; int N, K;
} mh_gemm_desc;
int mh_gemm_desc_launch(const mh_gemm_desc* d, mh_stream_t stream);
/* mh_gemm_dw_bias with both operands as K32 panels [cols / 32][ld rows][32] (panel != 0; M, N multiples of 32) */
int mh_gemm_dw_bias_ex(const void* A, int64_t lda, const void* B, int64_t ldb, int panel, float* out_partials, int splits, int64_t K, int M,
                       int N, int with_colsum, mh_stream_t stream);
/* mh_layernorm_bwd_drop whose second output (dx o keep / (1 - p)) has its own layout: row-major with pitch ldm or (m_panel) K32 panels
 * [H / 32][ldm rows][32]; always != 0: written even when the site drops nothing (a copy of dx in that layout) */
int mh_layernorm_bwd_ex(const void* x, const void* dy, const float* gamma, void* dx, void* dx_dropped, int64_t ldm, int m_panel, int always,
                        const mh_dropout* drop, float* partial, int n_partial, float* dgamma, float* dbeta, int accumulate, int64_t rows, int H,
                        float eps, int dtype, mh_stream_t stream);
/* mh_attention_stream_bwd_drop with O read from, and dq / dk / dv written as, K32 panels of the token-major tensors (o_panel / d_panel) */
int mh_attention_stream_bwd_layout(const void* q, const void* k, const void* v, const void* dO, const void* o, int o_panel, int64_t o_ld,
                                   const float* lse2, float* D, void* dq, void* dk, void* dv, int64_t ld_d, int d_panel, int B, int L, int nh,
                                   int dh, float scale, int64_t qkv_batch_stride, int64_t qkv_head_stride, int64_t qkv_row_stride,
                                   int64_t do_batch_stride, int64_t do_head_stride, int64_t do_row_stride, const uint32_t* keep_bits,
                                   float drop_p, mh_stream_t stream);
/* bf16 [rows, cols] row-major (pitch ld) <-> K32 panels [cols / 32][ld rows][32] (cols % 32 == 0); to_panel != 0: row-major in, panels out */
int mh_repack_panel(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int cols, int to_panel, mh_stream_t stream);

/* One encoder layer of the training step.  N = B L tokens; "panel" = bf16 K32 panels [cols / 32][ld][32], "rows" = bf16 row-major.
 * Weights: the bf16 working copies mh_weight_prep writes with its panel flags (mh_wprep_item.pad_ bits 0 / 1): w* = W [out][in] as panels
 * over `in`, w*_t = W^T [in][out] as panels over `out` (the input-gradient GEMMs' operand).  The forward keeps what the backward reads:
 * qkv rows [N][3H], vt (V^T in the streaming attention's key order, B nh dh L + 256 elements), ctx panel, lse [B nh L], pre1 / pre2
 * rows (pre-LayerNorm), x1 panel, g / dact panels [F / 32][N][32] (gelu and gelu' of the FFN pre-activation), keep_bits.
 * Dropout: p = 0 switches a site off; keep_bits (mh_dropout_bits_words(B nh, L) words) is read when bits_in != 0 (drawn ahead by
 * mh_dropout_bits), else written by the forward.
 * Backward: dy rows [N][H] in, dx rows out; `grads` fp32: dWqkv [3H][H] | dbqkv [3H] | dWao [H][H] | dbao [H] | dW1 [F][H] | db1 [F] |
 * dW2 [H][F] | db2 [H] | dln1_g | dln1_b | dln2_g | dln2_b (mh_train_layer_grad_floats in all), summation order fixed.  With side_stream the
 * folds of the split-K and LayerNorm partials run there, under the GEMMs that follow, and are joined into `stream` before the call returns
 * (one pair of events per device hands work over: the layers of one device are driven from one host thread at a time).
 * H = 512 (the full-row LayerNorm tile), F % 256 == 0, L % 64 == 0, L >= 512, head dim 32 or 64: mh_train_layer_supported. */
typedef struct mh_train_layer {
  int B, L, H, F, nh; float ln_eps;
  int64_t ld;   /* rows of every activation / gradient panel buffer (>= B L; not a power of two: panels 2^k bytes apart share their HBM channels) */
  const void *wqkv, *wqkv_t, *wao, *wao_t, *w1, *w1_t, *w2, *w2_t;
  const float *bqkv, *bao, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  mh_dropout drop_attn, drop_ao, drop_ffn;
  uint32_t* keep_bits; int bits_in;
  const void* x; int y_panel;
  void *qkv, *vt, *ctx; float* lse; void *pre1, *x1, *g, *dact, *pre2, *y;
  const void* dy; void* dx; void* scratch; size_t scratch_bytes; float* grads;
  mh_stream_t side_stream;   /* backward: a second stream of the same device for the gradient folds (NULL: everything on `stream`) */
} mh_train_layer;
int mh_train_layer_supported(int B, int L, int H, int F, int nh);
size_t mh_train_layer_scratch_bytes(int B, int L, int H, int F, int nh, int64_t ld);
int64_t mh_train_layer_grad_floats(int H, int F);
int mh_train_layer_fwd(const mh_train_layer* t, mh_stream_t stream);
int mh_train_layer_bwd(const mh_train_layer* t, mh_stream_t stream);

/* ---------------------------------------------------------------- per-launch timing (measurement)
 * Between mh_profile_start() and mh_profile_stop() every kernel this library launches is bracketed by two HIP events on its own
 * stream (not capturable: call outside hipGraph capture).  mh_profile_stop synchronises the device and writes one line per launch,
 * "kernel\tdetail\tgrid\tblock\tstream\tmilliseconds\n", into `out`; it returns the bytes the whole report needs.
 * The recorder is the only process-wide mutable state of libmusehip.so (everything else is per call or per device): one record list for the
 * whole process, guarded by a mutex - launches from several host threads may interleave while it is on; a launch made by thread A between
 * thread B's record and its kernel can attach to B's record, so time one thread at a time when the attribution matters. */
int mh_profile_start(void);
int64_t mh_profile_stop(char* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MUSEHIP_H */
