/* libmusehip_dbg.so ONLY (built with -DMH_ABLATE by `make dbg`): process-global A/B switches, ablation knobs and diagnostics of the
 * kernels.  None of these exists in the production library (libmusehip.so has no mutable global configuration: every default named
 * below is a compile-time constant there, tests/test_host_cpu.py checks the export lists of both builds).  They serve
 * tools/ (ab_step.py, ab_train.py, gemm_bench.py, attn_bench.py ...), `bench.py`'s A/B flags and the tests that compare an alternative
 * kernel form with the default one; several of them make outputs meaningless by design (timing-only ablations). */
#ifndef MUSEHIP_DBG_H
#define MUSEHIP_DBG_H
#include "musehip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A/B switch read by mh_denoiser_forward's panel path: 1 (default) = streaming kernel, 16-wave blocks with 256-key stages;
 * 2 = 8-wave blocks with 128-key stages (half a CU per block, same bits; 3% slower inside the step); 0 = resident / tiled kernels;
 * 3 = the in-kernel dropout-mask generator on the 16-wave geometry (spills; A/B); 4 = always the bound-checking build (by default
 * seq_len % 256 == 0 selects builds without the per-score bound compares, forward and backward). */
int mh_attention_set_stream(int on);

int mh_attention_set_variant(int resident);

/* Diagnostic: when non-NULL, the LDS-resident attention kernel writes 100 MHz timestamps per block into
 * stamps[block * 32 + {0: start, 1: K/V staged, 2 + w: wave w done}] (u64).  NULL (default) disables it. */
int mh_attention_set_profile(void* stamps);

/* A/B switch between the bf16 GEMM kernels: 0 = 128x128 tile, register-staged (the production library's fallback for shapes the big
 * tiles do not serve); 2 (default) = 256x128 tile (4 waves, 3-stage LDS-DMA ring, two blocks per CU; see mh_gemm_set_auto_wide);
 * 4 = 256x256 (8 waves, 4-stage ring) where N % 256 == 0.  (Rounds 1 - 5 carried more forms - 128x128 with global_load_lds, 256x256
 * ping-pong / four-wave, 256x128 on eight waves, bias-initialised accumulators, interleaved and staggered stage DMA: measured out and
 * removed in round 6; their numbers are in profiles/r0*_ab_nulls.txt and their code in the git history.) */
int mh_gemm_set_variant(int variant);

/* 1 (default): with the default variant, launches whose operands are all row-major (the training tape, direct callers) take the
 * 256 x 256 tile when N % 256 == 0 and it still gives every CU a block; K32-panel launches (the engine) never do.  0: round-2 rule. */
int mh_gemm_set_auto_wide(int on);

int mh_gemm_dw_set_blocks(int blocks);   /* A/B knob: blocks a launch aims for when choosing `splits` (default 512) */

/* A/B: 0 = the weight-gradient GEMM always on the 256 x 128 tile (4 waves, two blocks per CU: round 2), 1 (default) = a 256 x 256 tile on
 * 8 waves where N % 256 == 0: half the blocks, half the fp32 partials to write and fold.  mh_gemm_dw_splits follows the setting. */
int mh_gemm_dw_set_wide(int on);

/* A/B switch: mh_denoiser_forward's panel path uses mh_gemm_bias_res_ln for the two post-LN dense layers of an
 * encoder block when the hidden size allows it (default 1) or the separate GEMM + LayerNorm kernels (0). */
int mh_denoiser_set_fuse_ln(int on);

/* Timing-only ablation of the big-tile kernel (results are WRONG when non-zero): bit 0 skips the
 * DMA loads, bit 1 the MFMAs, bit 2 the epilogue stores, bit 3 the LDS fragment reads.  Used by tools/gemm_bench.py only. */
int mh_gemm_set_debug(int bits);

/* A/B mask of the big-tile GEMM epilogues / tiles.  Bit 0: QKV scatter with streaming (nt) q / k stores instead of ordinary ones
 * (default ordinary: attention reads them back at once; +0.9 % steps/s in round 2).  Bit 1: dense + GELU with ordinary instead of
 * streaming stores (default streaming: the output is large and read once; ordinary costs 3.8 % of the step).  The dense + residual +
 * LayerNorm epilogue always stores normally: its rows are re-read at once.  Bit 2: the full-row (LayerNorm) tile with the plain
 * instead of the ping-pong main loop (4 % slower step).  Bit 3 / 4: the 64-row full-row tile for K <= 512 / always (3 % / 7 % slower). */
int mh_gemm_set_plain_stores(int mask);

/* denoiser forward: 0 = never defer, 1 (default) = defer where the width has no full-row LayerNorm epilogue (d_model 768),
 * 2 = always (A/B) */
int mh_denoiser_set_defer_ln(int mode);

/* timing-only A/B knob (tools/ab_step.py skip): leave one kind of launch out of the bf16 panel forward (bit 0 QKV, 1 attention,
 * 2 attention-output dense + LN, 3 FFN1, 4 FFN2 + LN, 5 up-projection chain, 6 down-projection); outputs are then meaningless */
int mh_denoiser_set_skip(int mask);

/* panel LayerNorm kernels: 1 (default) = 4 rows per wave (four times the waves of the 16-row form), 0 = 16 rows per wave (A/B) */
int mh_layernorm_set_rows4(int on);

/* 1: the bf16 panel forward folds softmax scale x log2(e) into the stored queries and runs the pre-scaled attention (A/B; default 0) */
int mh_denoiser_set_prescale_q(int on);

/* timing-only ablation of the streaming kernel at head dim 64, seq_len <= 512 (tools/attn_bench.py): 1 no softmax vector work, 2 no S^T
 * MFMAs, 4 no P.V MFMAs, 8 no LDS fragment reads, 16 no stage DMA (built: 1 2 4 6 7 8 16 24 31); 0 = the real kernel */
int mh_attention_set_ablation(int bits);

/* A/B: 1 (default) = head and tail of the bf16 panel forward as one kernel each (mh_up_proj_ln_fused / mh_down_proj_fused), 0 = the
 * separate launches of rounds 1-3 */
int mh_denoiser_set_fuse_headtail(int on);

/* A/B: K32-panel launches (the sampler's engine) on the 256 x 256 tile by role - bit 0 dense + GELU (FFN1), bit 1 the QKV projection,
 * bit 2 every other launch whose N is a multiple of 256; 0 (default) = all of them on the 256 x 128 tile */
int mh_gemm_set_wide_roles(int mask);

/* A/B: 1 (default) = K32-panel launches of the big-tile kernels issue their stage DMA as `buffer_load_dwordx4 ... lds` (tile base in a descriptor,
 * K step as the scalar offset, a wave's pieces as immediate offsets: no vector address arithmetic per piece; bit-identical results);
 * 0 = global_load_lds with per-piece 64-bit addresses (rounds 1 - 4) */
int mh_gemm_set_buf_dma(int on);

/* A/B: 1 (default) = dense + bias + GELU of K32 panels into a K32-panel output (the sampler's FFN1: K = 512, M % 256 == 0, N % 128 == 0) runs
 * on the column-strip kernel (csrc/gemm_strip.h: mfma_32x32x16, two fragment sets, a ring that never drains, full-row stores through
 * v_permlane16_swap; bit-identical outputs); 0 = on gemm_big_kernel's 256 x 128 tile like every other shape (rounds 1 - 5) */
int mh_gemm_set_strip(int on);

/* Experiment (round 6, csrc/gemm_carry.h; result: profiles/r06_ffn1_carry.txt): out = gelu(A W^T + bias) for K32-panel A [K/32][lda][32],
 * W [K/32][ldw][32] and a K32-panel bf16 output [N/32][ldo][32] with the PREVIOUS tile's bias + GELU + store carried under the next tile's K
 * loop - one block of four waves per CU (one wave per SIMD, two accumulator sets), mfma_32x32x16, K loop unrolled.  K = 512, M % 2048 == 0,
 * N % 128 == 0.  variant (built: 0 1 2 5 7 8 11 12 14 16 17; csrc/gemm_carry.h lists them): 0 = carried, three-stage ring, 2 = six stages, 5 / 7 = the
 * epilogue after its own tile (one / two blocks per CU), 14 / 16 = 0 / 7 with whole-row stores (v_permlane16_swap), 12 = ordinary stores,
 * 1 / 8 / 11 / 17 = timing-only forms (main loop alone; no GELU).  Not bit-identical with the product kernel: another MFMA shape sums the K dimension in another order. */
int mh_gemm_ffn1_carry(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int64_t M, int N,
                       int K, int variant, mh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
