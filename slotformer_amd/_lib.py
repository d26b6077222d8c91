"""ctypes binding of libslotformer_hip.so (include/slotformer_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, a
RuntimeError is raised.  Nothing here imports oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SF_LIB_PATH') or os.path.join(_HERE, 'libslotformer_hip.so')   # (SF_LIB_PATH: a variant build, tools/ only)

FP = C.c_void_p  # device float*


class sf_tfm_layer(C.Structure):
    _fields_ = [(n, FP) for n in (
        'norm1_g', 'norm1_b', 'in_proj_w', 'in_proj_b', 'out_proj_w', 'out_proj_b',
        'norm2_g', 'norm2_b', 'lin1_w', 'lin1_b', 'lin2_w', 'lin2_b', 'lin1_packed', 'lin2_packed', 'attn_in_packed', 'attn_out_packed', 'tok_packed')]


class sf_rollouter(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        'num_slots', 'slot_size', 'd_model', 'num_layers', 'num_heads', 'ffn_dim', 'norm_first',
        'window_len', 'single_step')] + [(n, FP) for n in (
            'in_proj_w', 'in_proj_b', 'out_proj_w', 'out_proj_b', 'pe_tok')] + [
                ('layers', C.POINTER(sf_tfm_layer)), ('in_proj_packed', C.c_void_p), ('out_proj_packed', C.c_void_p)]


class sf_rollout_opts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('precision', 'seam_fused', 'ffn_rows', 'attn_heads_per_wg', 'attn_qkv_rows', 'ffn_tile', 'cus_available', 'layer_tok')]


class sf_tfm_layer_grads(C.Structure):
    _fields_ = [(n, FP) for n in (
        'norm1_g', 'norm1_b', 'in_proj_w', 'in_proj_b', 'out_proj_w', 'out_proj_b',
        'norm2_g', 'norm2_b', 'lin1_w', 'lin1_b', 'lin2_w', 'lin2_b')]


class sf_rollouter_grads(C.Structure):
    _fields_ = [(n, FP) for n in ('in_proj_w', 'in_proj_b', 'out_proj_w', 'out_proj_b')] + [
        ('layers', C.POINTER(sf_tfm_layer_grads)), ('pe_tok', FP)]


_SA_LEAVES = ('norm_in_g', 'norm_in_b', 'wk', 'wv', 'q_ln_g', 'q_ln_b', 'wq', 'gru_w_ih', 'gru_w_hh', 'gru_b_ih', 'gru_b_hh',
              'mlp_ln_g', 'mlp_ln_b', 'mlp_w1', 'mlp_b1', 'mlp_w2', 'mlp_b2')


class sf_slot_attention(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('in_features', 'slot_size', 'mlp_hidden', 'num_slots')] + [
        (n, FP) for n in _SA_LEAVES] + [('eps', C.c_float)]


class sf_slot_attention_grads(C.Structure):
    _fields_ = [(n, FP) for n in _SA_LEAVES]


class sf_savi_decoder_grads(C.Structure):
    _fields_ = [('deconv_w', FP * 8), ('deconv_b', FP * 8)] + [(n, FP) for n in ('out_w', 'out_b', 'pos_w', 'pos_b')]


class sf_savi_features(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('resolution', 'layers', 'channels', 'ks', 'hidden', 'out_channels')] + [
        ('conv_w', FP * 8), ('conv_b', FP * 8)] + [(n, FP) for n in ('pos_grid', 'pos_w', 'pos_b', 'ln_g', 'ln_b', 'fc1_w', 'fc1_b',
                                                                      'fc2_w', 'fc2_b')]


class sf_savi_features_grads(C.Structure):
    _fields_ = [('conv_w', FP * 8), ('conv_b', FP * 8)] + [(n, FP) for n in ('pos_w', 'pos_b', 'ln_g', 'ln_b', 'fc1_w', 'fc1_b',
                                                                              'fc2_w', 'fc2_b')]


class sf_savi_encoder(C.Structure):
    _fields_ = (
        [('resolution', C.c_int), ('enc_layers', C.c_int), ('enc_channels', C.c_int * 9),
         ('enc_ks', C.c_int)] +
        [(n, C.c_int) for n in ('enc_out_channels', 'num_slots', 'slot_size', 'slot_mlp_size',
                                'num_iterations')] +
        [('conv_w', FP * 8), ('conv_b', FP * 8), ('pos_table', FP)] +
        [(n, FP) for n in ('enc_ln_g', 'enc_ln_b', 'enc_fc1_w', 'enc_fc1_b', 'enc_fc2_w', 'enc_fc2_b',
                           'sa_norm_in_g', 'sa_norm_in_b', 'sa_q_ln_g', 'sa_q_ln_b', 'sa_q_w', 'sa_kv_w',
                           'gru_w_ih', 'gru_w_hh', 'gru_b_ih', 'gru_b_hh',
                           'mlp_ln_g', 'mlp_ln_b', 'mlp_w1', 'mlp_b1', 'mlp_w2', 'mlp_b2',
                           'init_latents')] +
        [('kd_mode', C.c_int)] +
        [(n, FP) for n in ('kd_w0', 'kd_b0', 'kd_ln_g', 'kd_ln_b', 'kd_w3', 'kd_b3')] +
        [(n, C.c_int) for n in ('pred_type', 'pred_rnn', 'pred_norm_first', 'pred_num_layers',
                                'pred_num_heads', 'pred_ffn_dim', 'pred_hidden')] +
        [(n, FP) for n in ('pm_ln_g', 'pm_ln_b', 'pm_w0', 'pm_b0', 'pm_w2', 'pm_b2')] +
        [('pred_layers', C.POINTER(sf_tfm_layer))] +
        [(n, FP) for n in ('lstm_w_ih', 'lstm_w_hh', 'lstm_b_ih', 'lstm_b_hh', 'proj_w', 'proj_b')] +
        [('sa_eps', C.c_float), ('sa_q_w_t', FP), ('pm_w0_t', FP), ('pm_w2_t', FP), ('kd_w0_t', FP),
         ('sa_gru_ih_p', C.c_void_p), ('sa_gru_hh_p', C.c_void_p), ('sa_mlp_w1_p', C.c_void_p), ('sa_mlp_w2_p', C.c_void_p),
         ('sa_q_w_p', C.c_void_p), ('sa_fold_q_w', FP), ('sa_fold_q_w_t', FP), ('sa_fold_gru_ih_t', FP), ('sa_fold_q_w_p', C.c_void_p),
         ('sa_fold_gru_ih_p', C.c_void_p), ('pred_packed', C.POINTER(C.c_void_p)), ('enc_fc1_p', C.c_void_p), ('enc_fc2_p', C.c_void_p),
         ('conv_w_frag', C.c_void_p * 8), ('pm_w0_p', C.c_void_p), ('pm_w2_p', C.c_void_p), ('kd_w0_p', C.c_void_p)])


class sf_slate_block(C.Structure):
    _fields_ = [(n, FP) for n in ('ln1_g', 'ln1_b', 'wqkv', 'wo', 'ln2_g', 'ln2_b', 'wq_c', 'wkv_c', 'wo_c', 'ln3_g', 'ln3_b',
                                  'w1', 'b1', 'w2', 'b2')] + [('is_first', C.c_int)]


class sf_slate_decoder(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('d_model', 'num_heads', 'num_layers', 'vocab_size', 'num_slots', 'max_len')] + [
        (n, FP) for n in ('in_proj_w', 'in_proj_b', 'tok_emb', 'pos_emb', 'lnf_g', 'lnf_b', 'head_w')] + [
            ('blocks', C.POINTER(sf_slate_block))]


class sf_savi_decoder(C.Structure):
    _fields_ = ([('resolution', C.c_int), ('dec_layers', C.c_int), ('dec_channels', C.c_int * 9),
                 ('dec_strides', C.c_int * 8), ('dec_ks', C.c_int), ('dec_res', C.c_int), ('num_slots', C.c_int),
                 ('slot_size', C.c_int), ('deconv_w', FP * 8), ('deconv_b', FP * 8), ('out_w', FP), ('out_b', FP),
                 ('pos_table', FP), ('deconv_w_flipped', FP * 8), ('deconv_w_frag', C.c_void_p * 8), ('l0_weff', FP), ('l0_posterm', FP)])


I, LL, F32, SZ, VP = C.c_int, C.c_longlong, C.c_float, C.c_size_t, C.c_void_p

# name -> (restype, argtypes): exactly the declarations of include/slotformer_hip.h
SIGNATURES = {
    'sf_version': (I, []),
    'sf_last_error_string': (C.c_char_p, []),
    'sf_get_precision': (I, []),
    'sf_set_precision': (I, [I]),
    'sf_profile_enable': (I, [I]),
    'sf_profile_sample': (I, [I]),
    'sf_set_seam_fused': (I, [I]),
    'sf_get_seam_fused': (I, []),
    'sf_set_ffn_rows64': (I, [I]),
    'sf_get_ffn_rows64': (I, []),
    'sf_profile_read': (I, [I, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    'sf_linear_f32': (I, [FP, I, FP, FP, FP, FP, F32, FP, I, FP, I, I, I, I, I, VP]),
    'sf_layernorm_f32': (I, [FP, FP, FP, FP, I, I, F32, VP]),
    'sf_conv2d_nchw_in_f32': (I, [FP, LL, FP, FP, FP, FP, I, I, I, I, I, I, I, I, VP]),
    'sf_conv2d_nhwc_f32': (I, [FP, FP, FP, FP, FP, I, I, I, I, I, I, I, VP]),
    'sf_pack_conv_weight_f32': (I, [FP, FP, I, I, I, VP]),
    'sf_pixel_feat_f32': (I, [FP] * 10 + [I, F32, I, VP]),
    'sf_conv_frag_bytes': (SZ, [I, I, I]),
    'sf_set_conv_fp16x2': (I, [I]),
    'sf_get_conv_fp16x2': (I, []),
    'sf_pack_conv_frag_weights': (I, [FP, VP, I, I, I, VP]),
    'sf_conv5x5_frag_f32': (I, [FP, VP, FP, FP, FP, I, I, I, I, VP]),
    'sf_conv5x5_ws_f32': (I, [FP, VP, FP, FP, FP, I, I, I, I, I, VP]),
    'sf_deconv_frag_bytes': (SZ, [I, I, I, I]),
    'sf_pack_deconv_frag_weights': (I, [FP, VP, I, I, I, I, VP]),
    'sf_deconv5x5s2_frag_f32': (I, [FP, VP, FP, FP, I, I, I, I, VP]),
    'sf_deconv5x5s2_head_frag_f32': (I, [FP, VP, FP, FP, FP, FP, I, I, I, VP]),
    'sf_decode_l0_expand_f32': (I, [FP, FP, FP, I, I, I, VP]),
    'sf_conv_transpose2d_nhwc_f32': (I, [FP, FP, FP, FP, I, I, I, I, I, I, I, I, VP]),
    'sf_pack_deconv_weight_f32': (I, [FP, FP, I, I, I, VP]),
    'sf_slot_broadcast_f32': (I, [FP, FP, FP, I, I, I, VP]),
    'sf_decode_combine_f32': (I, [FP, FP, FP, FP, I, I, I, VP]),
    'sf_savi_decode_workspace_bytes': (SZ, [C.POINTER(sf_savi_decoder), I]),
    'sf_savi_decode_f32': (I, [C.POINTER(sf_savi_decoder), FP, FP, FP, FP, I, VP, SZ, VP]),
    'sf_savi_decode_seg_f32': (I, [C.POINTER(sf_savi_decoder), FP, FP, FP, FP, VP, VP, F32, I, VP, SZ, VP]),
    'sf_decode_combine_seg_f32': (I, [FP, FP, FP, FP, VP, VP, F32, VP, I, I, I, VP]),
    'sf_postproc_mask_f32': (I, [FP, VP, VP, F32, VP, I, I, I, VP]),
    'sf_pos_embed_table_f32': (I, [FP, FP, FP, FP, I, I, VP]),
    'sf_slot_attn_num_partials': (I, [I]),
    'sf_slot_attn_iter_f32': (I, [FP, FP, I, LL, FP, FP, FP, FP, I, I, I, I, F32, F32, VP]),
    'sf_slot_update_f32': (I, [FP, FP, I, FP] + [FP] * 10 + [FP, I, I, I, I, F32, VP]),
    'sf_slot_update_packed_f32': (I, [FP, FP, I, FP, VP, VP, FP, FP, FP, FP, VP, FP, VP, FP, FP, FP, FP, VP, FP, I, I, I, I, F32, VP]),
    'sf_mha_f32': (I, [FP, FP, I, I, I, I, I, VP]),
    'sf_qkv_attention_f32': (I, [FP, FP, FP, F32, FP, FP, FP, I, I, I, I, I, VP]),
    'sf_lstm_pointwise_f32': (I, [FP, FP, FP, FP, I, I, VP]),
    'sf_sample_dist_f32': (I, [FP, FP, FP, I, I, VP]),
    'sf_bilinear_resize_f32': (I, [FP, FP, LL, I, I, I, I, VP]),
    'sf_rollout_workspace_bytes': (SZ, [C.POINTER(sf_rollouter), I]),
    'sf_rollout_f32': (I, [C.POINTER(sf_rollouter), FP, I, I, I, VP, SZ, VP]),
    'sf_rollout_opts_f32': (I, [C.POINTER(sf_rollouter), FP, I, I, I, VP, SZ, VP, C.POINTER(sf_rollout_opts)]),
    'sf_rollout_uses_seam': (I, [C.POINTER(sf_rollouter), I]),
    'sf_rollout_uses_seam_opts': (I, [C.POINTER(sf_rollouter), I, C.POINTER(sf_rollout_opts)]),
    'sf_rollout_is_fused': (I, [C.POINTER(sf_rollouter)]),
    'sf_rollout_tok_ok': (I, [C.POINTER(sf_rollouter)]),
    'sf_ffn_chunk_partials_f32': (I, [C.POINTER(sf_tfm_layer), FP, LL, FP, LL, I, I, I, VP]),
    'sf_attn_block_f32': (I, [C.POINTER(sf_tfm_layer), FP, FP, I, I, I, I, VP]),
    'sf_ffn_block_rows_f32': (I, [C.POINTER(sf_tfm_layer), FP, FP, I, I, VP]),
    'sf_set_layer_tok': (I, [I]),
    'sf_get_layer_tok': (I, []),
    'sf_layer_tok_packed_bytes': (SZ, []),
    'sf_pack_layer_tok_weights': (I, [C.POINTER(sf_tfm_layer), VP, I, I, I, VP]),
    'sf_layer_tok_block_f32': (I, [C.POINTER(sf_tfm_layer), I, FP, FP, I, I, VP]),
    'sf_debug_read_ts_layer_tok': (I, [C.POINTER(C.c_longlong)]),
    'sf_attn_rows_planes_bytes': (SZ, [I]),
    'sf_attn_block_rows_f32': (I, [C.POINTER(sf_tfm_layer), FP, FP, VP, I, I, I, VP]),
    'sf_slot_attn_iter_bwd_workspace_bytes': (SZ, [I, I, I, I]),
    'sf_slot_attn_iter_bwd_f32': (I, [FP, FP, I, LL, FP, FP, FP, I, FP, FP, FP, I, FP, I, I, I, I, F32, F32, VP, SZ, VP]),
    'sf_slot_attention_train_workspace_bytes': (SZ, [C.POINTER(sf_slot_attention), I, I, I]),
    'sf_slot_attention_train_fwd_f32': (I, [C.POINTER(sf_slot_attention), FP, FP, I, I, I, FP, VP, SZ, VP]),
    'sf_slot_attention_train_bwd_f32': (I, [C.POINTER(sf_slot_attention), FP, FP, FP, FP, C.POINTER(sf_slot_attention_grads), I, I, I,
                                            VP, SZ, VP]),
    'sf_savi_decode_train_workspace_bytes': (SZ, [C.POINTER(sf_savi_decoder), I]),
    'sf_savi_decode_train_fwd_f32': (I, [C.POINTER(sf_savi_decoder), FP, FP, FP, FP, I, VP, SZ, VP]),
    'sf_savi_decode_train_bwd_f32': (I, [C.POINTER(sf_savi_decoder), C.POINTER(C.c_void_p), FP, FP, FP, C.POINTER(sf_savi_decoder_grads),
                                         I, VP, SZ, VP]),
    'sf_savi_features_train_workspace_bytes': (SZ, [C.POINTER(sf_savi_features), I]),
    'sf_savi_features_train_fwd_f32': (I, [C.POINTER(sf_savi_features), FP, LL, I, FP, VP, SZ, VP]),
    'sf_savi_features_train_bwd_f32': (I, [C.POINTER(sf_savi_features), FP, LL, FP, C.POINTER(sf_savi_features_grads), I, VP, SZ, VP]),
    'sf_linear_bwd_workspace_bytes': (SZ, [LL, I, I]),
    'sf_linear_bwd_f32': (I, [FP, FP, FP, FP, FP, FP, FP, LL, I, I, I, VP, SZ, VP]),
    'sf_mha_train_fwd_f32': (I, [FP, FP, I, I, I, I, F32, C.c_ulonglong, VP]),
    'sf_mha_train_bwd_f32': (I, [FP, FP, FP, I, I, I, I, F32, C.c_ulonglong, VP]),
    'sf_dropout_f32': (I, [FP, FP, FP, LL, F32, C.c_ulonglong, VP]),
    'sf_adam_flat_f32': (I, [FP, FP, FP, FP, LL, I, F32, F32, F32, F32, VP]),
    'sf_layernorm_bwd_workspace_bytes': (SZ, [I]),
    'sf_layernorm_bwd_f32': (I, [FP, FP, FP, FP, FP, FP, LL, I, F32, VP, SZ, VP]),
    'sf_rollout_train_workspace_bytes': (SZ, [C.POINTER(sf_rollouter), I, I]),
    'sf_rollout_train_fwd_f32': (I, [C.POINTER(sf_rollouter), FP, FP, I, I, F32, C.c_ulonglong, VP, SZ, VP]),
    'sf_rollout_train_bwd_f32': (I, [C.POINTER(sf_rollouter), FP, FP, C.POINTER(sf_rollouter_grads), I, I, F32,
                                     C.c_ulonglong, VP, SZ, VP]),
    'sf_savi_encode_workspace_bytes': (SZ, [C.POINTER(sf_savi_encoder), I]),
    'sf_savi_encode_f32': (I, [C.POINTER(sf_savi_encoder), FP, FP, FP, FP, FP, I, FP, FP, FP, I, I, VP, SZ,
                               VP]),
    'sf_groupnorm1_workspace_bytes': (SZ, [I]),
    'sf_groupnorm1_nhwc_f32': (I, [FP, FP, FP, FP, I, I, I, I, F32, I, I, VP, SZ, VP]),
    'sf_groupnorm1_bwd_workspace_bytes': (SZ, [I, I, I, I]),
    'sf_groupnorm1_nhwc_bwd_f32': (I, [FP, FP, FP, FP, FP, FP, FP, I, I, I, I, F32, I, I, VP, SZ, VP]),
    'sf_slate_attention_f32': (I, [FP, FP, FP, FP, I, I, I, I, I, I, I, I, I, I, VP]),
    'sf_slate_attention_strided_f32': (I, [FP, FP, FP, FP, I, I, I, I, LL, LL, LL, LL, I, I, I, I, I, I, VP]),
    'sf_slate_attention_bwd_workspace_bytes': (SZ, [I, I, I]),
    'sf_slate_attention_bwd_f32': (I, [FP, FP, FP, FP, FP, FP, FP, FP, I, I, I, I, LL, LL, LL, LL, I, I, I, I, I, I, VP, SZ, VP]),
    'sf_slate_attention_train_fwd_f32': (I, [FP, FP, FP, FP, FP, I, I, I, I, LL, LL, LL, LL, I, I, I, I, I, I, F32, C.c_ulonglong, VP]),
    'sf_slate_attention_train_bwd_f32': (I, [FP, FP, FP, FP, FP, FP, FP, FP, FP, I, I, I, I, LL, LL, LL, LL, I, I, I, I, I, I, F32,
                                             C.c_ulonglong, VP, SZ, VP]),
    'sf_embed_tokens_f32': (I, [VP, FP, FP, FP, I, I, I, VP]),
    'sf_argmax_rows_f32': (I, [FP, LL, VP, LL, I, VP]),
    'sf_cross_entropy_f32': (I, [FP, VP, FP, FP, LL, I, VP]),
    'sf_softmax_rows_f32': (I, [FP, FP, F32, FP, LL, I, VP]),
    'sf_softmax_rows_bwd_f32': (I, [FP, FP, F32, FP, LL, I, VP]),
    'sf_log_softmax_rows_f32': (I, [FP, FP, LL, I, VP]),
    'sf_cross_entropy_bwd_f32': (I, [FP, VP, FP, FP, LL, I, VP]),
    'sf_gumbel_softmax_rows_f32': (I, [FP, C.c_ulonglong, F32, FP, LL, I, VP]),
    'sf_slate_generate_workspace_bytes': (SZ, [C.POINTER(sf_slate_decoder), I, I]),
    'sf_slate_generate_f32': (I, [C.POINTER(sf_slate_decoder), FP, I, I, VP, FP, VP, SZ, VP]),
    'sf_packed_linear_bytes': (SZ, [I, I]),
    'sf_pack_linear_weights': (I, [FP, VP, I, I, VP]),
    'sf_attn_packed_bytes': (SZ, [I, I]),
    'sf_pack_attn_weights': (I, [FP, FP, VP, VP, I, I, VP]),
    'sf_ffn_packed_bytes': (SZ, [I, I]),
    'sf_pack_ffn_weights': (I, [FP, FP, VP, VP, I, I, VP]),
    'sf_savi_cnn_workspace_bytes': (SZ, [C.POINTER(sf_savi_encoder), I]),
    'sf_savi_cnn_f32': (I, [C.POINTER(sf_savi_encoder), FP, I, I, I, I, FP, VP, SZ, VP]),
    'sf_savi_encode_pre_f32': (I, [C.POINTER(sf_savi_encoder), FP, FP, I, FP, FP, FP, FP, I, FP, FP, FP, I, I, VP, SZ,
                                   VP]),
    'sf_savi_encode_fork_workspace_bytes': (SZ, [C.POINTER(sf_savi_encoder), I, I]),
    'sf_savi_encode_batched_workspace_bytes': (SZ, [C.POINTER(sf_savi_encoder), I, I]),
    'sf_set_encode_fuse_next': (I, [I]),
    'sf_get_encode_fuse_next': (I, []),
    'sf_set_slot_chain': (I, [I]),
    'sf_get_slot_chain': (I, []),
    'sf_set_slot_attn_planes': (I, [I]),
    'sf_get_slot_attn_planes': (I, []),
    'sf_set_pixel_tok': (I, [I]),
    'sf_get_pixel_tok': (I, []),
    'sf_savi_chain_ok': (I, [C.POINTER(sf_savi_encoder), I, I]),
    'sf_savi_planes_bytes': (SZ, [C.POINTER(sf_savi_encoder), I, I]),
    'sf_savi_features_workspace_bytes': (SZ, [C.POINTER(sf_savi_encoder), I, I]),
    'sf_savi_features_planes_f32': (I, [C.POINTER(sf_savi_encoder), FP, I, I, VP, VP, SZ, VP]),
    'sf_savi_slots_chain_workspace_bytes': (SZ, [C.POINTER(sf_savi_encoder), I]),
    'sf_savi_slots_chain_f32': (I, [C.POINTER(sf_savi_encoder), VP, FP, FP, FP, LL, FP, FP, I, I, I, VP, SZ, VP]),
    'sf_savi_encode_fork_f32': (I, [C.POINTER(sf_savi_encoder), FP, FP, I, FP, FP, FP, FP, I, FP, FP, FP, I, I, VP, SZ,
                                    VP, VP]),
    'sf_kv_producer_workspace_bytes': (SZ, [I, I]),
    'sf_kv_producer_f32': (I, [FP] * 11 + [I, I, I, I, F32, VP, SZ, VP]),
    # device-memory helpers + host-buffer twins (same signatures as the device entry points)
    'sf_device_alloc': (I, [C.POINTER(VP), SZ]),
    'sf_device_free': (I, [VP]),
    'sf_device_upload': (I, [VP, VP, SZ]),
    'sf_device_download': (I, [VP, VP, SZ]),
    'sf_device_synchronize': (I, []),
    'sf_stream_create_cu_mask': (I, [C.POINTER(VP), C.POINTER(C.c_uint), I]),
    'sf_stream_cus': (I, [VP]),
    'sf_stream_set_cus': (I, [VP, I]),
    'sf_debug_spin': (I, [I, VP]),
    'sf_debug_clock_probe': (I, [I, VP, VP]),
    'sf_stream_destroy': (I, [VP]),
    'sf_slot_attn_iter_f32_host': (I, [FP, FP, I, LL, FP, FP, FP, FP, I, I, I, I, F32, F32, VP]),
    'sf_rollout_f32_host': (I, [C.POINTER(sf_rollouter), FP, I, I, I, VP, SZ, VP]),
    'sf_savi_encode_f32_host': (I, [C.POINTER(sf_savi_encoder), FP, FP, FP, FP, FP, I, FP, FP, FP, I, I, VP, SZ,
                                    VP]),
    'sf_savi_decode_f32_host': (I, [C.POINTER(sf_savi_decoder), FP, FP, FP, FP, I, VP, SZ, VP]),
    'sf_seam_timeouts': (I, []),
    'sf_slot_attn_iter_bf16': (I, [VP, VP, I, LL, FP, FP, FP, FP, I, I, I, I, F32, F32, VP]),
    'sf_rollout_bf16': (I, [C.POINTER(sf_rollouter), FP, I, I, I, VP, SZ, VP]),
    'sf_rollout_bf16_host': (I, [C.POINTER(sf_rollouter), FP, I, I, I, VP, SZ, VP]),
    'sf_slot_update_f32_host': (I, [FP, FP, I, FP] + [FP] * 10 + [FP, I, I, I, I, F32, VP]),
    'sf_kv_producer_f32_host': (I, [FP] * 11 + [I, I, I, I, F32, VP, SZ, VP]),
}

_lib = None


class _CaptureGate:
    """Stream capture and kernel launches from other host threads do not mix on this runtime: with a hipGraph capture under way on one thread
    (thread-local capture mode) launches from other threads were seen to return wrong results, `invalid argument`, or to crash hipStreamEndCapture
    (tests/test_rollout_opts_gpu.py::test_options_are_per_call_and_per_thread late in a long process).  So every call into the library holds this
    gate SHARED and a capture (EncodeRolloutPipeline building its graphs) holds it EXCLUSIVE: launches of other threads wait the few milliseconds a
    capture takes; the capturing thread's own calls pass."""

    def __init__(self):
        import threading
        self._cond = threading.Condition()
        self._readers = 0
        self._writer = None
        self._depth = 0
        self._ident = threading.get_ident

    def enter_shared(self):
        me = self._ident()
        with self._cond:
            if self._writer == me:
                return False
            while self._writer is not None:
                self._cond.wait()
            self._readers += 1
            return True

    def exit_shared(self):
        with self._cond:
            self._readers -= 1
            if self._readers == 0:
                self._cond.notify_all()

    def __enter__(self):
        me = self._ident()
        with self._cond:
            if self._writer == me:
                self._depth += 1
                return self
            while self._writer is not None or self._readers > 0:
                self._cond.wait()
            self._writer, self._depth = me, 1
        return self

    def __exit__(self, *exc):
        with self._cond:
            self._depth -= 1
            if self._depth == 0:
                self._writer = None
                self._cond.notify_all()
        return False


CAPTURE_GATE = _CaptureGate()


class _Gated:
    """The loaded library with every entry point behind CAPTURE_GATE (shared)."""

    def __init__(self, h):
        self.__dict__['_h'] = h

    def __getattr__(self, name):
        f = getattr(self._h, name)
        if not callable(f):
            return f
        gate = CAPTURE_GATE

        def call(*a):
            held = gate.enter_shared()
            try:
                return f(*a)
            finally:
                if held:
                    gate.exit_shared()
        call.__name__ = name
        self.__dict__[name] = call
        return call


def lib():
    """Load the HIP library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: the HIP extension is not built. Run '
                '`python -m slotformer_amd.build` (or __graft_entry__.build()). There is no fallback path.')
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f'cannot load {LIB_PATH}: {e}') from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = _Gated(h)
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().sf_last_error_string().decode(errors='replace')
        raise RuntimeError(f'libslotformer_hip: {msg}')
