"""ctypes binding of libmusehip.so (C ABI: include/musehip.h).

There is no fallback: `lib()` raises if the shared library is absent or does not export the
symbols the header declares.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C musediffusion_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree build.  Another build is taken ONLY under the explicit A/B switch of tools/ab_lib.sh
# (MUSEHIP_AB=1 MUSEHIP_LIB=<path>): a stray MUSEHIP_LIB alone changes nothing
LIB_PATH = os.path.join(_HERE, "csrc", "libmusehip.so")
if os.environ.get("MUSEHIP_AB") == "1" and os.environ.get("MUSEHIP_LIB"):
    LIB_PATH = os.environ["MUSEHIP_LIB"]

MH_F32, MH_BF16, MH_BF16X3, MH_F16X3 = 0, 1, 2, 3
ACT_NONE, ACT_TANH, ACT_GELU_ERF, ACT_SILU = 0, 1, 2, 3

c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)
VP = C.c_void_p
I64 = C.c_int64
INT = C.c_int
F32 = C.c_float


class StepCoef(C.Structure):
    """mh_step_coef"""
    _fields_ = [("coef1", F32), ("coef2", F32), ("sigma", F32), ("recip", F32), ("recipm1", F32),
                ("sqrt_abp", F32), ("dir", F32), ("pad", F32)]


class LoopState(C.Structure):
    """mh_loop_state"""
    _fields_ = [("pos", C.c_uint32), ("n_steps", C.c_uint32), ("cur_t", C.c_int32), ("rng_step", C.c_uint32)]


class OptHParams(C.Structure):
    """mh_opt_hparams"""
    _fields_ = [("beta1", F32), ("beta2", F32), ("eps", F32), ("one_minus_beta1", F32), ("one_minus_beta2", F32),
                ("decay_mul", F32), ("step_size", F32), ("bias2_sqrt", F32), ("n_ema", INT), ("ema_rate", F32 * 4),
                ("ema_one_minus", F32 * 4)]


class Dropout(C.Structure):
    """mh_dropout"""
    _fields_ = [("p", F32), ("seed", C.c_uint64), ("offset", C.c_uint64), ("mask", VP)]


class StepRng(C.Structure):
    """mh_step_rng"""
    _fields_ = [("seed", C.c_uint64), ("stream_id", C.c_uint32), ("bound", F32), ("step_counter", VP), ("first_elem", C.c_int64)]


class StepUpdate(C.Structure):
    """mh_step_update"""
    _fields_ = [("x", VP), ("x_start", VP), ("mask", VP), ("mask_per_elem", INT), ("table", VP), ("coef", VP), ("clip", INT), ("ddim", INT),
                ("pred_xstart", VP), ("mean_out", VP), ("noise", VP), ("rng", C.POINTER(StepRng))]


class WPrepItem(C.Structure):
    """mh_wprep_item"""
    _fields_ = [("src", VP), ("dst", VP), ("dst_t", VP), ("rows", C.c_int32), ("cols", C.c_int32), ("ld_dst", C.c_int64),
                ("ld_t", C.c_int64), ("tile_start", C.c_int32), ("pad_", C.c_int32)]


class GemmDesc(C.Structure):
    """mh_gemm_desc"""
    _fields_ = [("A", VP), ("lda", I64), ("a_panel", INT), ("W", VP), ("ldw", I64), ("w_panel", INT), ("bias", VP),
                ("residual", VP), ("ldr", I64), ("r_panel", INT), ("out", VP), ("ldo", I64), ("o_panel", INT), ("out_f32", INT),
                ("pre_out", VP), ("ldp", I64), ("p_panel", INT), ("pre_kind", INT), ("act", INT), ("act_grad", INT),
                ("ln_gamma", VP), ("ln_beta", VP), ("ln_eps", F32), ("drop", C.POINTER(Dropout)), ("M", I64), ("N", INT), ("K", INT)]


class TrainLayer(C.Structure):
    """mh_train_layer"""
    _fields_ = ([(n, INT) for n in ("B", "L", "H", "F", "nh")] + [("ln_eps", F32), ("ld", I64)] +
                [(n, VP) for n in ("wqkv", "wqkv_t", "wao", "wao_t", "w1", "w1_t", "w2", "w2_t")] +
                [(n, VP) for n in ("bqkv", "bao", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b")] +
                [("drop_attn", Dropout), ("drop_ao", Dropout), ("drop_ffn", Dropout), ("keep_bits", VP), ("bits_in", INT),
                 ("x", VP), ("y_panel", INT)] +
                [(n, VP) for n in ("qkv", "vt", "ctx", "lse", "pre1", "x1", "g", "dact", "pre2", "y")] +
                [("dy", VP), ("dx", VP), ("scratch", VP), ("scratch_bytes", C.c_size_t), ("grads", VP), ("side_stream", VP)])


class LayerWeights(C.Structure):
    """mh_layer_weights"""
    _fields_ = [(n, VP) for n in ("w_qkv", "b_qkv", "w_ao", "b_ao", "ln1_g", "ln1_b", "w_ff1", "b_ff1",
                                  "w_ff2", "b_ff2", "ln2_g", "ln2_b", "w_qkv_f", "c1_qkv", "c2_qkv", "w_ff1_f", "c1_ff1", "c2_ff1")]


class LnDefer(C.Structure):
    """mh_ln_defer"""
    _fields_ = [("a_stats", VP), ("a_slots", INT), ("c1", VP), ("r_stats", VP), ("r_slots", INT), ("r_gamma", VP), ("r_beta", VP),
                ("o_stats", VP), ("o_slots", INT), ("h_norm", INT), ("eps", F32)]


class Denoiser(C.Structure):
    """mh_denoiser"""
    _fields_ = ([("dtype", INT)] +
                [(n, INT) for n in ("E", "H", "F", "nh", "nL", "Tt", "Tt_pad", "T4_pad", "E_pad", "L_max")] +
                [("panel", INT), ("has_proj", INT), ("ln_eps", F32)] +
                [(n, VP) for n in ("w_t0", "b_t0", "w_t2", "b_t2", "w_up0", "b_up0", "w_up2", "b_up2", "pos",
                                   "ln0_g", "ln0_b", "w_dn0", "b_dn0", "w_dn2", "b_dn2")] +
                [("layers", C.POINTER(LayerWeights))])


# name -> (restype, argtypes); every symbol include/musehip.h declares
SIGNATURES = {
    "mh_last_error": (C.c_char_p, []),
    "mh_abi_version": (INT, []),
    "mh_device_name": (INT, [INT, C.c_char_p, INT]),
    "mh_cast_pad": (INT, [VP, I64, VP, I64, I64, I64, I64, INT, VP]),
    "mh_cast_to_f32": (INT, [VP, I64, VP, I64, I64, I64, INT, VP]),
    "mh_pack_panel": (INT, [VP, I64, VP, I64, I64, I64, I64, VP]),
    "mh_unpack_panel_f32": (INT, [VP, I64, VP, I64, I64, I64, VP]),
    "mh_split_supported": (INT, [INT]),
    "mh_split_pack": (INT, [VP, I64, VP, I64, I64, INT, INT, INT, VP]),
    "mh_split_join": (INT, [VP, I64, VP, I64, I64, INT, INT, INT, VP]),
    "mh_split_layernorm": (INT, [VP, I64, VP, VP, VP, VP, VP, VP, I64, I64, INT, INT, F32, INT, VP]),
    "mh_split_gemm": (INT, [VP, I64, VP, I64, VP, INT, VP, I64, VP, I64, INT, I64, I64, INT, INT, INT, INT, VP]),
    "mh_split_gemm_res_ln_supported": (INT, [INT]),
    "mh_split_gemm_res_ln": (INT, [VP, I64, VP, I64, VP, VP, I64, VP, VP, F32, VP, I64, I64, INT, INT, INT, VP]),
    "mh_split_attention": (INT, [VP, I64, INT, I64, VP, I64, I64, VP, I64, INT, INT, INT, INT, F32, INT, VP]),
    "mh_row_sqnorm": (INT, [VP, VP, INT, INT, VP]),
    "mh_embed_gather": (INT, [VP, VP, VP, I64, INT, INT, VP]),
    "mh_timestep_embedding": (INT, [VP, VP, INT, INT, I64, F32, INT, VP]),
    "mh_gemm_bias_act": (INT, [VP, I64, VP, I64, VP, VP, I64, VP, I64, INT, I64, INT, INT, INT, INT, VP]),
    "mh_gemm_bias_act_ex": (INT, [VP, I64, INT, VP, I64, INT, VP, VP, I64, INT, VP, I64, INT, INT, I64, INT, INT, INT, INT, VP]),
    "mh_gemm_batched": (INT, [VP, I64, I64, VP, I64, I64, VP, VP, I64, I64, INT, INT, I64, INT, INT, INT, VP]),
    "mh_gemm_qkv": (INT, [VP, I64, VP, I64, VP, VP, VP, VP, INT, INT, INT, INT, INT, VP]),
    "mh_gemm_qkv_ex": (INT, [VP, I64, INT, VP, I64, INT, VP, VP, VP, VP, INT, INT, INT, INT, INT, VP]),
    "mh_attention_fwd_ex": (INT, [VP, VP, VP, VP, I64, INT, INT, INT, INT, INT, F32, INT, VP]),
    "mh_attention_stream_fwd": (INT, [VP, VP, VP, VP, I64, INT, INT, INT, INT, INT, F32, VP]),
    "mh_attention_stream_supported": (INT, [INT, INT]),
    "mh_attention_stream_fwd_lse": (INT, [VP, VP, VP, VP, I64, INT, INT, INT, INT, INT, F32, VP, VP]),
    "mh_attention_stream_bwd": (INT, [VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, I64, INT, INT, INT, INT, F32, VP]),
    "mh_attention_stream_fwd_ex": (INT, [VP, VP, VP, VP, I64, INT, INT, INT, INT, INT, F32, VP, I64, I64, I64, VP]),
    "mh_attention_stream_bwd_ex": (INT, [VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, I64, INT, INT, INT, INT, F32, I64, I64, I64, I64, I64, I64, VP]),
    "mh_attention_stream_bwd_supported": (INT, [INT, INT]),
    "mh_attention_bwd_rowdot": (INT, [VP, VP, I64, VP, INT, INT, INT, INT, VP]),
    "mh_attention_stream_enabled": (INT, []),
    "mh_gemm_qkv_vtperm": (INT, [VP, I64, INT, VP, I64, INT, VP, VP, VP, VP, INT, INT, INT, INT, VP]),
    "mh_attention_fwd": (INT, [VP, VP, VP, VP, I64, INT, INT, INT, INT, F32, INT, VP]),
    "mh_layernorm": (INT, [VP, VP, VP, VP, I64, INT, F32, INT, VP]),
    "mh_layernorm_panel": (INT, [VP, I64, VP, VP, VP, I64, I64, INT, F32, VP]),
    "mh_add_pos_time_layernorm_panel": (INT, [VP, I64, INT, VP, VP, VP, VP, VP, VP, I64, INT, INT, INT, F32, VP]),
    "mh_add_pos_time_layernorm": (INT, [VP, I64, INT, VP, VP, VP, VP, VP, VP, INT, INT, INT, F32, INT, VP]),
    "mh_round_to_embedding": (INT, [VP, VP, VP, VP, I64, INT, INT, VP]),
    "mh_round_workspace_bytes": (C.c_size_t, [I64, INT, INT]),
    "mh_round_to_embedding_mfma": (INT, [VP, VP, VP, VP, I64, INT, INT, VP, C.c_size_t, VP]),
    "mh_logits_argmax": (INT, [VP, VP, VP, VP, I64, INT, INT, VP]),
    "mh_q_sample": (INT, [VP, VP, VP, VP, VP, INT, VP, INT, I64, INT, VP]),
    "mh_p_sample_epilogue": (INT, [VP, VP, VP, VP, VP, VP, INT, INT, VP, INT, VP, VP, VP, VP, INT, I64, INT, VP]),
    "mh_ddim_epilogue": (INT, [VP, VP, VP, VP, VP, VP, INT, INT, VP, INT, VP, VP, VP, INT, I64, INT, VP]),
    "mh_trunc_normal": (INT, [VP, I64, F32, C.c_uint64, C.c_uint32, VP, VP]),
    "mh_trunc_normal_at": (INT, [VP, I64, I64, F32, C.c_uint64, C.c_uint32, VP, VP]),
    "mh_stream_delay": (C.c_int, [C.c_uint, VP]),
    "mh_transpose": (INT, [VP, I64, I64, VP, I64, I64, INT, INT, INT, INT, VP]),
    "mh_head_permute": (INT, [VP, VP, I64, INT, INT, INT, INT, INT, INT, VP]),
    "mh_col_sum": (INT, [VP, I64, I64, INT, INT, I64, VP, INT, VP, INT, INT, VP]),
    "mh_add_pos_time": (INT, [VP, I64, VP, VP, VP, INT, INT, INT, INT, VP]),
    "mh_act_fwd": (INT, [VP, VP, I64, INT, INT, VP]),
    "mh_act_bwd": (INT, [VP, VP, VP, I64, INT, INT, VP]),
    "mh_layernorm_bwd": (INT, [VP, VP, VP, VP, VP, INT, VP, VP, INT, I64, INT, F32, INT, VP]),
    "mh_layernorm_bwd_drop": (INT, [VP, VP, VP, VP, VP, VP, VP, INT, VP, VP, INT, I64, INT, F32, INT, VP]),
    "mh_softmax_rows": (INT, [VP, I64, INT, I64, F32, INT, VP]),
    "mh_softmax_bwd_rows": (INT, [VP, VP, I64, INT, I64, F32, INT, VP]),
    "mh_cross_entropy_fwd": (INT, [VP, I64, VP, VP, VP, I64, INT, VP]),
    "mh_cross_entropy_bwd": (INT, [VP, I64, VP, VP, VP, VP, I64, I64, INT, INT, INT, VP]),
    "mh_sqdiff_mean": (INT, [VP, VP, F32, VP, INT, I64, VP]),
    "mh_sqdiff_bwd": (INT, [VP, VP, F32, VP, VP, VP, INT, INT, I64, VP]),
    "mh_add_inplace": (INT, [VP, VP, I64, INT, VP]),
    "mh_scatter_add_rows": (INT, [VP, VP, VP, I64, INT, INT, VP, C.c_size_t, VP]),
    "mh_scatter_add_rows_workspace_bytes": (C.c_size_t, [INT, INT]),
    "mh_sum_slices": (INT, [VP, INT, I64, VP, VP]),
    "mh_batch_max_row": (INT, []),
    "mh_ragged_to_padded": (INT, [VP, VP, VP, VP, INT, INT, C.c_int32, VP]),
    "mh_meta_to_batch": (INT, [VP, INT, VP, VP, INT, INT, VP]),
    "mh_corrupt_masking_token": (INT, [VP, VP, VP, F32, VP, INT, VP]),
    "mh_corrupt_masking_note": (INT, [VP, VP, VP, F32, VP, INT, VP]),
    "mh_corrupt_randomize_note": (INT, [VP, VP, VP, VP, F32, VP, INT, VP]),
    "mh_corrupt_random_rotating": (INT, [VP, VP, VP, INT, VP, VP, INT, VP]),
    "mh_controllability_counts": (INT, [VP, VP, VP, INT, VP, INT, INT, VP]),
    "mh_msim_vectors": (INT, [VP, VP, VP, VP, INT, INT, F32, VP]),
    "mh_validate_tokens": (INT, [VP, VP, VP, INT, INT, VP]),
    "mh_scale_rows": (INT, [VP, VP, VP, VP, INT, INT, I64, INT, VP]),
    "mh_adamw_ema_step": (INT, [VP, VP, INT, C.POINTER(OptHParams), VP]),
    "mh_grad_norm": (INT, [VP, VP, INT, VP, VP, VP]),
    "mh_clip_grads": (INT, [VP, VP, INT, VP, F32, VP]),
    "mh_step_begin": (INT, [VP, VP, VP, VP, VP, INT, VP]),
    "mh_step_end": (INT, [VP, VP]),
    "mh_gemm_dw": (INT, [VP, I64, VP, I64, VP, INT, I64, INT, INT, VP]),
    "mh_gemm_dw_splits": (INT, [I64, INT, INT]),
    "mh_weight_prep": (INT, [VP, INT, INT, VP]),
    "mh_distance_scores": (INT, [VP, I64, VP, VP, VP, I64, I64, INT, VP]),
    "mh_gemm_desc_launch": (INT, [C.POINTER(GemmDesc), VP]),
    "mh_gemm_dw_bias_ex": (INT, [VP, I64, VP, I64, INT, VP, INT, I64, INT, INT, INT, VP]),
    "mh_layernorm_bwd_ex": (INT, [VP, VP, VP, VP, VP, I64, INT, INT, VP, VP, INT, VP, VP, INT, I64, INT, F32, INT, VP]),
    "mh_attention_stream_bwd_layout": (INT, [VP, VP, VP, VP, VP, INT, I64, VP, VP, VP, VP, VP, I64, INT, INT, INT, INT, INT, F32, I64, I64, I64, I64, I64, I64, VP, F32, VP]),
    "mh_repack_panel": (INT, [VP, I64, VP, I64, I64, INT, INT, VP]),
    "mh_train_layer_supported": (INT, [INT, INT, INT, INT, INT]),
    "mh_train_layer_scratch_bytes": (C.c_size_t, [INT, INT, INT, INT, INT, I64]),
    "mh_train_layer_grad_floats": (I64, [INT, INT]),
    "mh_train_layer_fwd": (INT, [C.POINTER(TrainLayer), VP]),
    "mh_train_layer_bwd": (INT, [C.POINTER(TrainLayer), VP]),
    "mh_gemm_dw_bias": (INT, [VP, I64, VP, I64, VP, INT, I64, INT, INT, INT, VP]),
    "mh_gemm_act_grad": (INT, [VP, I64, VP, I64, VP, I64, VP, I64, I64, INT, INT, INT, VP]),
    "mh_gemm_bias_act_pre": (INT, [VP, I64, VP, I64, VP, VP, VP, I64, I64, INT, INT, INT, VP]),
    "mh_gemm_bias_act_dact": (INT, [VP, I64, VP, I64, VP, VP, VP, I64, I64, INT, INT, INT, VP]),
    "mh_gemm_bias_res_ln": (INT, [VP, I64, INT, VP, I64, INT, VP, VP, I64, INT, VP, VP, F32, VP, I64, INT, I64, INT, INT, VP]),
    "mh_gemm_bias_res_ln_supported": (INT, [INT]),
    "mh_denoiser_get_fuse_ln": (INT, []),
    "mh_graph_begin_capture": (INT, [VP]),
    "mh_graph_end_capture": (INT, [VP, C.POINTER(VP)]),
    "mh_graph_launch": (INT, [VP, VP]),
    "mh_graph_destroy": (INT, [VP]),
    "mh_dropout_fwd": (INT, [VP, I64, VP, I64, I64, INT, INT, C.POINTER(Dropout), VP]),
    "mh_gemm_bias_dropout_res": (INT, [VP, I64, VP, I64, VP, VP, I64, VP, I64, I64, INT, INT, INT, C.POINTER(Dropout), VP]),
    "mh_gemm_bias_dropout_res_ln": (INT, [VP, I64, VP, I64, VP, VP, I64, VP, VP, F32, VP, VP, I64, I64, INT, INT, C.POINTER(Dropout), VP]),
    "mh_dropout_bits_words": (C.c_size_t, [INT, INT]),
    "mh_dropout_bits": (INT, [VP, INT, INT, C.POINTER(Dropout), VP]),
    "mh_dropout_bits_apply": (INT, [VP, I64, VP, INT, INT, F32, INT, VP]),
    "mh_attention_stream_fwd_drop": (INT, [VP, VP, VP, VP, I64, INT, INT, INT, INT, INT, F32, VP, I64, I64, I64, C.POINTER(Dropout), VP, INT, VP]),
    "mh_attention_stream_bwd_drop": (INT, [VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, I64, INT, INT, INT, INT, F32, I64, I64, I64, I64, I64, I64, VP, F32, VP]),
    "mh_gemm_bias_act_defer": (INT, [VP, I64, VP, I64, VP, VP, I64, VP, I64, I64, INT, INT, INT, C.POINTER(LnDefer), VP]),
    "mh_gemm_qkv_vtperm_defer": (INT, [VP, I64, VP, I64, VP, VP, VP, VP, INT, INT, INT, INT, C.POINTER(LnDefer), VP]),
    "mh_denoiser_get_defer_ln": (INT, []),
    "mh_gemm_qkv_vtperm_qs": (INT, [VP, I64, VP, I64, VP, VP, VP, VP, INT, INT, INT, INT, F32, C.POINTER(LnDefer), VP]),
    "mh_attention_stream_prescaled_supported": (INT, [INT, INT]),
    "mh_attention_stream_fwd_prescaled": (INT, [VP, VP, VP, VP, I64, INT, INT, INT, INT, INT, VP]),
    "mh_profile_start": (INT, []),
    "mh_profile_stop": (I64, [C.c_char_p, C.c_size_t]),
    "mh_denoiser_workspace_bytes": (C.c_size_t, [C.POINTER(Denoiser), INT, INT]),
    "mh_time_embed": (INT, [C.POINTER(Denoiser), VP, VP, INT, VP, C.c_size_t, VP]),
    "mh_denoiser_forward": (INT, [C.POINTER(Denoiser), VP, VP, VP, VP, INT, INT, VP, C.c_size_t, VP]),
    "mh_up_proj_ln_fused_supported": (INT, [INT, INT, INT]),
    "mh_down_proj_fused_supported": (INT, [INT, INT]),
    "mh_up_proj_ln_fused": (INT, [VP, INT, INT, VP, VP, VP, VP, VP, VP, VP, VP, VP, F32, VP, I64, INT, INT, INT, VP]),
    "mh_down_proj_fused": (INT, [VP, I64, VP, VP, VP, VP, VP, VP, I64, INT, INT, VP]),
    "mh_denoiser_gives_sqnorm": (INT, [C.POINTER(Denoiser)]),
    "mh_denoiser_forward_sqnorm": (INT, [C.POINTER(Denoiser), VP, VP, VP, VP, VP, INT, INT, VP, C.c_size_t, VP]),
    "mh_down_proj_round_supported": (INT, [INT, INT, INT]),
    "mh_round_split_bytes": (C.c_size_t, [INT, INT]),
    "mh_round_split_table": (INT, [VP, VP, INT, INT, VP, VP]),
    "mh_down_proj_round_fused": (INT, [VP, I64, VP, VP, VP, VP, VP, VP, VP, INT, VP, VP, I64, INT, INT, VP]),
    "mh_denoiser_rounds_in_forward": (INT, [C.POINTER(Denoiser), INT]),
    "mh_denoiser_forward_round": (INT, [C.POINTER(Denoiser), VP, VP, VP, VP, VP, INT, VP, VP, INT, INT, VP, C.c_size_t, VP]),
    "mh_round_slots": (INT, [INT]),
    "mh_round_scores": (INT, [VP, VP, VP, VP, VP, VP, I64, INT, INT, VP]),
    "mh_step_epilogue_slots": (INT, [INT, VP, VP, VP, VP, INT, VP, VP, INT, INT, VP, INT, VP, VP, VP, VP, VP, VP, INT, I64, INT, VP]),
    "mh_step_advance": (INT, [VP, VP, VP, VP, VP, INT, VP]),
    "mh_denoiser_phases_supported": (INT, [C.POINTER(Denoiser)]),
    "mh_denoiser_head": (INT, [C.POINTER(Denoiser), VP, VP, VP, VP, I64, INT, INT, VP, C.c_size_t, VP]),
    "mh_denoiser_layers": (INT, [C.POINTER(Denoiser), VP, I64, VP, I64, INT, INT, VP, C.c_size_t, VP]),
    "mh_denoiser_tail": (INT, [C.POINTER(Denoiser), VP, I64, VP, INT, INT, VP, C.c_size_t, VP]),
}

# include/musehip_dbg.h: exported by libmusehip_dbg.so only (A/B switches, ablation knobs, diagnostics)
DBG_SIGNATURES = {
    "mh_gemm_set_wide_roles": (INT, [INT]),
    "mh_denoiser_set_fuse_headtail": (INT, [INT]),
    "mh_attention_set_stream": (INT, [INT]),
    "mh_attention_set_variant": (INT, [INT]),
    "mh_attention_set_profile": (INT, [VP]),
    "mh_gemm_set_variant": (INT, [INT]),
    "mh_gemm_set_auto_wide": (INT, [INT]),
    "mh_gemm_dw_set_blocks": (INT, [INT]),
    "mh_gemm_dw_set_wide": (INT, [INT]),
    "mh_denoiser_set_fuse_ln": (INT, [INT]),
    "mh_gemm_set_debug": (INT, [INT]),
    "mh_gemm_set_plain_stores": (INT, [INT]),
    "mh_denoiser_set_defer_ln": (INT, [INT]),
    "mh_denoiser_set_skip": (INT, [INT]),
    "mh_layernorm_set_rows4": (INT, [INT]),
    "mh_denoiser_set_prescale_q": (INT, [INT]),
    "mh_attention_set_ablation": (INT, [INT]),
    "mh_gemm_set_buf_dma": (INT, [INT]),
    "mh_gemm_set_strip": (INT, [INT]),
    "mh_gemm_ffn1_carry": (INT, [VP, I64, VP, I64, VP, VP, I64, I64, INT, INT, INT, VP]),
}

_lib = None
_handles = {}


class MuseHipError(RuntimeError):
    pass


DBG_LIB_PATH = os.path.join(_HERE, "csrc", "libmusehip_dbg.so")


def _load(path, signatures, what):
    if path in _handles:
        return _handles[path]
    if not os.path.exists(path):
        raise MuseHipError(
            "%s not found at %s: the HIP kernels are the only compute path "
            "(build with `make -C musediffusion_amd/csrc` or __graft_entry__.build())" % (what, path))
    # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Import torch
    # FIRST so the library's DT_NEEDED entry resolves to that already-loaded runtime: two HIP
    # runtimes in one process do not share devices, streams or allocations.
    import torch  # noqa: F401
    handle = C.CDLL(path)
    for name, (res, args) in signatures.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise MuseHipError("%s does not export %s (stale build?)" % (what, name))
        fn.restype = res
        fn.argtypes = args
    _handles[path] = handle
    return handle


def lib():
    """The loaded library (cached).  Raises MuseHipError when it cannot be loaded.  This is the production build unless the caller
    switched to the debug build (use_debug_library / debug_library)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, SIGNATURES, "libmusehip.so")
    return _lib


def use_debug_library(on=True):
    """Route every later call of this process through libmusehip_dbg.so (-DMH_ABLATE: the same kernels plus the switches of
    include/musehip_dbg.h) or back to the production library.  For tools/, bench.py's A/B flags and tests that compare kernel forms;
    the product never calls this.  Each library has its own switch state; objects made under one (captured graphs, engines' arenas)
    are plain HIP objects and stay valid under the other."""
    global _lib
    _lib = _load(DBG_LIB_PATH, {**SIGNATURES, **DBG_SIGNATURES}, "libmusehip_dbg.so") if on else _load(LIB_PATH, SIGNATURES, "libmusehip.so")
    return _lib


class debug_library:
    """`with _lib.debug_library(): ...` - the debug build for the duration of the block."""

    def __enter__(self):
        self.prev = _lib
        return use_debug_library(True)

    def __exit__(self, *a):
        global _lib
        _lib = self.prev


def check(rc, what=""):
    if rc != 0:
        msg = lib().mh_last_error()
        raise MuseHipError("%s failed (%d): %s" % (what or "libmusehip call", rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MuseHipError("musediffusion_amd runs on the GPU only: got a %s tensor "
                               "(move the model and inputs to cuda; there is no CPU path)" % t.device)
