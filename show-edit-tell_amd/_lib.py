"""ctypes binding of the C-ABI library csrc/libset_hip.so (include/set_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails this module
raises.  PyTorch is used only for device memory and streams; every tensor crosses the boundary
as a raw device pointer + sizes.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SET_LIB_VARIANT=exp: the experimental build of the same sources (build.py LIB_EXP; bench.py's `experimental` leg and the
# split-precision parity test run in child processes with it).  Anything else: the shipped library.
LIB_PATH = os.path.join(_HERE, "csrc", "libset_hip_exp.so" if os.environ.get("SET_LIB_VARIANT") == "exp" else "libset_hip.so")

SET_OK = 0
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3

c_f32p = C.c_void_p
c_i64p = C.c_void_p


class SetError(RuntimeError):
    pass


class EditNetDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "T", "R", "F", "D", "A", "V", "maxT", "adaptive")]


EDITNET_WEIGHT_FIELDS = (
    # (struct field, state_dict key)
    ("embed", "embed.embedding.weight"),
    ("enc_x2h_w", "caption_encoder.lstm_encoder_cell.x2h.weight"),
    ("enc_x2h_b", "caption_encoder.lstm_encoder_cell.x2h.bias"),
    ("enc_h2h_w", "caption_encoder.lstm_encoder_cell.h2h.weight"),
    ("enc_h2h_b", "caption_encoder.lstm_encoder_cell.h2h.bias"),
    ("enc_aff_w", "caption_encoder.affine_hn.weight"),
    ("enc_aff_b", "caption_encoder.affine_hn.bias"),
    ("ca_feat_w", "caption_attention.cap_features_att.weight"),
    ("ca_feat_b", "caption_attention.cap_features_att.bias"),
    ("ca_dec_w", "caption_attention.cap_decoder_att.weight"),
    ("ca_dec_b", "caption_attention.cap_decoder_att.bias"),
    ("ca_full_w", "caption_attention.cap_full_att.weight"),
    ("ca_full_b", "caption_attention.cap_full_att.bias"),
    ("ca_gate_w", "caption_attention.context_gate.weight"),
    ("ca_gate_b", "caption_attention.context_gate.bias"),
    ("ca_sc_w", "caption_attention.sc_affine.weight"),
    ("ca_sc_b", "caption_attention.sc_affine.bias"),
    ("ca_tc_w", "caption_attention.tc_affine.weight"),
    ("ca_tc_b", "caption_attention.tc_affine.bias"),
    ("va_emb_w", "visual_attention.att_embed.0.weight"),
    ("va_emb_b", "visual_attention.att_embed.0.bias"),
    ("va_feat_w", "visual_attention.features_att.weight"),
    ("va_feat_b", "visual_attention.features_att.bias"),
    ("va_dec_w", "visual_attention.decoder_att.weight"),
    ("va_dec_b", "visual_attention.decoder_att.bias"),
    ("va_full_w", "visual_attention.full_att.weight"),
    ("va_full_b", "visual_attention.full_att.bias"),
    ("al_wih", "attention_lstm.weight_ih"),
    ("al_whh", "attention_lstm.weight_hh"),
    ("al_bih", "attention_lstm.bias_ih"),
    ("al_bhh", "attention_lstm.bias_hh"),
    ("cl_x2h_w", "copy_lstm.x2h.weight"),
    ("cl_x2h_b", "copy_lstm.x2h.bias"),
    ("cl_h2h_w", "copy_lstm.h2h.weight"),
    ("cl_h2h_b", "copy_lstm.h2h.bias"),
    ("cl_cnew_w", "copy_lstm.gate_cnew.weight"),
    ("cl_cnew_b", "copy_lstm.gate_cnew.bias"),
    ("cl_cmem_w", "copy_lstm.gate_cmem.weight"),
    ("cl_cmem_b", "copy_lstm.gate_cmem.bias"),
    ("fc_w", "fc.weight"),
    ("fc_b", "fc.bias"),
)


class EditNetWeights(C.Structure):
    # the state_dict tensors + the optional derived token table (include/set_hip.h)
    _fields_ = [(f, C.c_void_p) for f, _ in EDITNET_WEIGHT_FIELDS] + [("tok_table", C.c_void_p)]


class ProfileEntry(C.Structure):
    _fields_ = [("tag", C.c_char * 32), ("launches", C.c_int), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


def profile_report(max_entries=64):
    """[{tag, launches, ms, flops, bytes}] aggregated since set_profile_enable(1)."""
    lib = load()
    arr = (ProfileEntry * max_entries)()
    n = lib.set_profile_report(arr, max_entries)
    if n < 0:
        check(-n, "set_profile_report")
    return [dict(tag=arr[i].tag.decode(), launches=arr[i].launches, ms=arr[i].ms, flops=arr[i].flops,
                 bytes=arr[i].bytes) for i in range(n)]


class DcnetDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "T", "D", "A", "C", "E", "V", "maxT")]


DCNET_WEIGHT_FIELDS = (
    ("embed", "embed.embedding.weight"),
    ("enc_wih_f", "caption_encoder.lstm_encoder.weight_ih_l0"),
    ("enc_whh_f", "caption_encoder.lstm_encoder.weight_hh_l0"),
    ("enc_bih_f", "caption_encoder.lstm_encoder.bias_ih_l0"),
    ("enc_bhh_f", "caption_encoder.lstm_encoder.bias_hh_l0"),
    ("enc_wih_b", "caption_encoder.lstm_encoder.weight_ih_l0_reverse"),
    ("enc_whh_b", "caption_encoder.lstm_encoder.weight_hh_l0_reverse"),
    ("enc_bih_b", "caption_encoder.lstm_encoder.bias_ih_l0_reverse"),
    ("enc_bhh_b", "caption_encoder.lstm_encoder.bias_hh_l0_reverse"),
    ("enc_cat_w", "caption_encoder.concat.weight"),
    ("enc_cat_b", "caption_encoder.concat.bias"),
    ("ca_feat_w", "caption_attention.cap_features_att.weight"),
    ("ca_feat_b", "caption_attention.cap_features_att.bias"),
    ("ca_dec_w", "caption_attention.cap_decoder_att.weight"),
    ("ca_dec_b", "caption_attention.cap_decoder_att.bias"),
    ("ca_full_w", "caption_attention.cap_full_att.weight"),
    ("ca_full_b", "caption_attention.cap_full_att.bias"),
    ("al_wih", "attention_lstm.weight_ih"),
    ("al_whh", "attention_lstm.weight_hh"),
    ("al_bih", "attention_lstm.bias_ih"),
    ("al_bhh", "attention_lstm.bias_hh"),
    ("ll_wih", "language_lstm.weight_ih"),
    ("ll_whh", "language_lstm.weight_hh"),
    ("ll_bih", "language_lstm.bias_ih"),
    ("ll_bhh", "language_lstm.bias_hh"),
    ("fc_w", "fc.weight"),
    ("fc_b", "fc.bias"),
)


class DcnetWeights(C.Structure):
    _fields_ = [(f, C.c_void_p) for f, _ in DCNET_WEIGHT_FIELDS] + [("tok_table", C.c_void_p)]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_longlong), ("B", C.c_void_p), ("ldb", C.c_longlong),
                ("C", C.c_void_p), ("ldc", C.c_longlong), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("accumulate", C.c_int)]


class SlabSrc(C.Structure):
    """include/set_hip.h SetSlabSrc: one addend of a gradient that is still split-K partials"""
    _fields_ = [("p", C.c_void_p), ("slab_stride", C.c_int64), ("ld", C.c_int64), ("nslab", C.c_int32), ("rows", C.c_int32)]


class ColsumDesc(C.Structure):
    """include/set_hip.h SetColsumDesc: one column sum of a grouped launch (set_colsum_group_f32)"""
    _fields_ = [("x", C.c_void_p), ("ld", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32), ("out", C.c_void_p),
                ("accumulate", C.c_int32), ("out2", C.c_void_p), ("accumulate2", C.c_int32)]


def _fields(spec):
    """'int a, b; float c; ptr d, e' style field list -> ctypes _fields_"""
    kinds = {"int": C.c_int, "float": C.c_float, "u64": C.c_uint64, "i64": C.c_int64, "ptr": C.c_void_p, "size": C.c_size_t}
    out = []
    for part in spec.split(";"):
        part = part.strip()
        if not part:
            continue
        kind, names = part.split(None, 1)
        for nme in names.split(","):
            nme = nme.strip()
            if "[" in nme:
                base, cnt = nme[:-1].split("[")
                out.append((base, kinds[kind] * int(cnt)))
            else:
                out.append((nme, kinds[kind]))
    return out


class XELoopArgs(C.Structure):
    """include/set_hip.h SetXELoopArgs"""
    _fields_ = _fields("int T, B, R, F, Tc, D, A, V, train; float p_embed, p_out; u64 seed, off_embed, off_out; ptr bts; ptr w; "
                       "ptr E, al_wih, al_whh; ptr tok; i64 tok_step, tok_stride; ptr X, H, Mem, mask, att1_c, pre1, att1; "
                       "i64 att1_step; ptr EMB, H1, C1, H2, C2, G1, G2, WHC, ZT, S, TT, ALPHAC, ALPHAV, ATT2C, ATT2V, SEL, CNEW, CG, "
                       "X2, H2D; ptr gated, cx, aimg; ptr ws_l; size ws_l_bytes; ptr ws_c; size ws_c_bytes; ptr ws_k; size ws_k_bytes; "
                       "int step_logs; ptr rmask; i64 rmask_step")


class XEBwdLoopArgs(C.Structure):
    """include/set_hip.h SetXEBwdLoopArgs"""
    _fields_ = _fields("int T, B, R, F, Tc, D, A, acc_datt1; float p_out; u64 seed, off_out; ptr bts; "
                       "ptr cl_cnew_w, cl_cmem_w, cl_x2h_w, cl_h2h_w, w_ctx, w_h1, dec_cat, al_wih, al_whh, va_full, ca_full; "
                       "ptr G1, G2, C1, C2, CG, SEL, CNEW, ZT, S, TT, ALPHAC, ALPHAV, ATT2C, ATT2V, X, H, Mem, att1_c, att1; "
                       "i64 att1_step; ptr dH2D; ptr DU, DGW, DSZT, DATT2, DWFC, DWFV, DEC, DEV, DCTX, DG1, datt1; i64 datt1_step; "
                       "ptr datt1c, dMem; ptr DC1[2], DC2[2]; ptr dcm, dcn, dop, dalc; ptr slab_ws[5]; size slab_ws_bytes; ptr tmp[11]; "
                       "ptr DLAST")


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_Z = C.c_size_t
_U = C.c_uint64

# name -> (restype, argtypes); mirrors include/set_hip.h one to one
PROTOTYPES = {
    "set_abi_version": (_I, []),
    "set_error_string": (C.c_char_p, [_I]),
    "set_last_hip_error": (_I, []),
    "set_last_hip_error_string": (C.c_char_p, []),
    "set_target_arch": (C.c_char_p, []),
    "set_profile_enable": (_I, [_I]),
    "set_profile_report": (_I, [_P, _I]),
    "set_editnet_workspace_bytes": (_Z, [C.POINTER(EditNetDims)]),
    "set_editnet_token_table_bytes": (_Z, [C.POINTER(EditNetDims)]),
    "set_editnet_token_table_workspace_bytes": (_Z, [C.POINTER(EditNetDims)]),
    "set_editnet_build_token_table": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _P, _Z, _P]),
    "set_editnet_begin": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _P, _P, _P, _P, _Z, _P]),
    "set_editnet_step": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _P, _L, _I, _P, _L, _P, _Z, _P]),
    "set_editnet_greedy_pick": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _L, _I, _L, _P, _P, _I,
                                     _P, _Z, _P]),
    "set_editnet_greedy": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _P, _P, _P, _L, _L, _I, _P, _P,
                                _P, _Z, _P]),
    "set_decode_row_limits": (_I, [_P]),
    "set_editnet_greedy_begun": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _L, _L, _I, _P, _P, _P, _Z, _P]),
    "set_editnet_beam_persistent": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _P, _P, _P, _L, _L, _I, _P, _P, _P, _P, _P,
                                         _P, _Z, _P]),
    "set_editnet_sample": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _P, _P, _P, _L, _L, _I, _U, _U, _P,
                                _P, _P, _Z, _P]),
    "set_editnet_xe_forward": (_I, [C.POINTER(EditNetWeights), C.POINTER(EditNetDims), _P, _P, _P, _L,
                                    C.POINTER(C.c_int), _P, _P, _P, _P, _Z, _P]),
    "set_editnet_ws_tensor": (_P, [C.POINTER(EditNetDims), _P, C.c_char_p]),
    "set_dcnet_workspace_bytes": (_Z, [C.POINTER(DcnetDims)]),
    "set_dcnet_token_table_bytes": (_Z, [C.POINTER(DcnetDims)]),
    "set_dcnet_token_table_workspace_bytes": (_Z, [C.POINTER(DcnetDims)]),
    "set_dcnet_build_token_table": (_I, [C.POINTER(DcnetWeights), C.POINTER(DcnetDims), _P, _P, _Z, _P]),
    "set_dcnet_begin": (_I, [C.POINTER(DcnetWeights), C.POINTER(DcnetDims), _P, _P, _P, _Z, _P]),
    "set_dcnet_step": (_I, [C.POINTER(DcnetWeights), C.POINTER(DcnetDims), _P, _L, _I, _P, _L, _P, _Z, _P]),
    "set_dcnet_greedy_pick": (_I, [C.POINTER(DcnetWeights), C.POINTER(DcnetDims), _P, _L, _I, _L, _P, _P, _I, _P, _Z,
                                   _P]),
    "set_dcnet_greedy": (_I, [C.POINTER(DcnetWeights), C.POINTER(DcnetDims), _P, _P, _L, _L, _I, _P, _P, _P, _Z, _P]),
    "set_dcnet_sample": (_I, [C.POINTER(DcnetWeights), C.POINTER(DcnetDims), _P, _P, _L, _L, _I, _U, _U, _P, _P, _P, _Z,
                              _P]),
    "set_dcnet_xe_forward": (_I, [C.POINTER(DcnetWeights), C.POINTER(DcnetDims), _P, _L, C.POINTER(C.c_int), _P, _P,
                                  _P, _P, _Z, _P]),
    "set_dcnet_ws_tensor": (_P, [C.POINTER(DcnetDims), _P, C.c_char_p]),
    "set_linear_workspace_bytes": (_Z, [_I, _I, _I]),
    "set_linear_f32": (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _P, _Z, _P]),
    "set_embed_relu_f32": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, _P]),
    "set_lstm_cell_workspace_bytes": (_Z, [_I, _I, _I]),
    "set_lstm_cell_f32": (_I, [_P, _L, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _I, _P, _Z, _P]),
    "set_caption_attention_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "set_caption_attention_f32": (_I, [C.POINTER(EditNetWeights), _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P,
                                       _Z, _P]),
    "set_visual_attention_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "set_visual_attention_f32": (_I, [C.POINTER(EditNetWeights), _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z,
                                      _P]),
    "set_visual_attention_masked_f32": (_I, [C.POINTER(EditNetWeights), _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P,
                                             _Z, _P]),
    "set_select_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "set_select_soft_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "set_select_soft_bwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "set_copy_lstm_workspace_bytes": (_Z, [_I, _I, _I]),
    "set_copy_lstm_f32": (_I, [C.POINTER(EditNetWeights), _P, _L, _I, _P, _P, _P, _P, _P, _I, _I, _P, _Z, _P]),
    "set_lstm_cell_train_f32": (_I, [_P, _L, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _P, _Z, _P]),
    "set_lstm_cell_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "set_copy_lstm_train_f32": (_I, [C.POINTER(EditNetWeights), _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P,
                                     _Z, _P]),
    "set_copy_gate_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "set_lstm_gates_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "set_caption_attention_train_f32": (_I, [C.POINTER(EditNetWeights), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I,
                                             _I, _I, _I, _I, _P, _Z, _P]),
    "set_context_gate_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "set_attention_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "set_encoder_cell_workspace_bytes": (_Z, [_I, _I]),
    "set_encoder_cell_train_f32": (_I, [_P, _L, _L, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _L, _L, _I, _P, _I, _I, _P,
                                        _Z, _P]),
    "set_encoder_cell_bwd_f32": (_I, [_P, _P, _P, _P, _L, _L, _I, _P, _I, _P, _P, _P, _P, _L, _P, _P, _I, _I, _P]),
    "set_select_bwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "set_editnet_attentions_workspace_bytes": (_Z, [_I, _I, _I]),
    "set_editnet_attentions_train_f32": (_I, [C.POINTER(EditNetWeights)] + [_P] * 20 + [_I] * 6 + [_P, _Z, _P]),
    "set_copy_gate_bwd_ld_f32": (_I, [_P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "set_caption_attention_att2_f32": (_I, [C.POINTER(EditNetWeights), _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P,
                                            _Z, _P]),
    "set_context_gate_bwd_ld_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "set_lstm_cell_pre_train_f32": (_I, [_P, _L, _P, _L, _I, _P, _L, _P, _L, _I, _P, _P, _P, _L, _P, _P, _P, _P, _I, _I, _P, _Z,
                                         _P]),
    "set_select_bwd_acc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "set_attention_bwd_acc_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _P]),
    "set_attention_dvalues_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "set_xe_loss_f32": (_I, [_P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _P, _P, _P, _P]),
    "set_xe_loss_bwd_f32": (_I, [_P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _P, _P, _P, _L, _P]),
    "set_clip_adam_workspace_bytes": (_Z, [_I, _P]),
    "set_clip_adam_f32": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _I, _P, _P, _Z, _P]),
    "set_colsum_workspace_bytes": (_Z, [_I]),
    "set_colsum_f32": (_I, [_P, _L, _I, _I, _P, _I, _P, _Z, _P]),
    "set_colsum_group_workspace_bytes": (_Z, [_P, _I]),
    "set_colsum_group_f32": (_I, [_P, _I, _P, _Z, _P]),
    "set_dropout_f32": (_I, [_P, _L, _P, _L, _I, _I, C.c_float, _U, _U, _P]),
    "set_embed_relu_dropout_f32": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, C.c_float, _U, _U, _P]),
    "set_dropout_bwd_f32": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, C.c_float, _I, _P]),
    "set_dropout_steps_f32": (_I, [_P, _L, _P, _L, _L, _I, _I, _I, C.c_float, _U, _U, _P]),
    "set_dropout_bwd_steps_f32": (_I, [_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, C.c_float, _I, _P]),
    "set_dropout_bwd_philox_f32": (_I, [_P, _L, _P, _L, _I, _I, C.c_float, _U, _U, _I, _P]),
    "set_rowsum_mask_f32": (_I, [_P, _L, _I, _I, _P, _P]),
    "set_pack_f32": (_I, [_P, _L, _I, _I, C.POINTER(_P), C.POINTER(_L), C.POINTER(_I), _I, _P]),
    "set_sample_pick_f32": (_I, [_P, _L, _I, _I, _I, _I, _L, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P]),
    "set_sample_logp_bwd_f32": (_I, [_P, _L, _P, _P, _P, _P, _L, _I, _I, _P]),
    "set_philox4x32": (_I, [_P, _I, _U, _U, _P]),
    "set_beam_pick_f32": (_I, [_P, _P, _L, _I, _I, _I, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "set_beam_gather_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "set_gemm_group_f32": (_I, [C.POINTER(GemmDesc), _I, _I, _I, _P, _Z, _P]),
    "set_gemm_group_slabs_f32": (_I, [C.POINTER(GemmDesc), _I, _I, _I, _P, _Z, C.POINTER(SlabSrc), _P]),
    "set_editnet_xe_train_loop_f32": (_I, [C.POINTER(XELoopArgs), _P]),
    "set_editnet_xe_train_bwd_loop_f32": (_I, [C.POINTER(XEBwdLoopArgs), _P]),
    "set_lstm_cell_bwd_src_f32": (_I, [C.POINTER(SlabSrc), _I, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "set_copy_gate_bwd_src_f32": (_I, [C.POINTER(SlabSrc), _I, _P, _L, C.c_float, _U, _U, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P,
                                       _I, _I, _P]),
    "set_lstm_gates_bwd_src_f32": (_I, [_P, C.POINTER(SlabSrc), _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "set_select_bwd_src_f32": (_I, [_P, C.POINTER(SlabSrc), _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "set_context_gate_bwd_src_f32": (_I, [_P, C.POINTER(SlabSrc), _I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "set_attention_bwd_src_f32": (_I, [_P, C.POINTER(SlabSrc), _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                       _L, _P]),
    "set_gemm_f32": (_I, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _I, _P, _Z, _P]),
    "set_ciderd_create": (_P, [_P, _P, _P, _L, C.c_double, _I, C.c_double]),
    "set_ciderd_destroy": (None, [_P]),
    "set_ciderd_score": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "set_caption_encoder_workspace_bytes": (_Z, [_I, _I, _I]),
    "set_caption_encoder_f32": (_I, [C.POINTER(EditNetWeights), _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
}

_lib = None


def load():
    """Load libset_hip.so (once).  Raises SetError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SetError(
            "HIP library %s is missing: build it with `python -m show_edit_tell_amd.build` "
            "(there is no CPU fallback for the decode path)" % LIB_PATH)
    # torch bundles its own HIP runtime (same SONAME as /opt/rocm's): import it FIRST so that this
    # library binds to the runtime that owns torch's device pointers and streams.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            MISSING.append(name)         # header / library drift: calling it raises below; tests assert none
            setattr(lib, name, _missing(name))
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


MISSING = []


def _missing(name):
    def fn(*a, **k):
        raise SetError("libset_hip.so does not export %s (rebuild: python -m show_edit_tell_amd.build --force)" % name)
    return fn


def check(rc: int, what: str = ""):
    if rc == SET_OK:
        return
    lib = load()
    msg = lib.set_error_string(rc).decode()
    if rc == 3:
        msg += ": " + lib.set_last_hip_error_string().decode()
    raise SetError("%s failed (code %d): %s" % (what or "libset_hip call", rc, msg))


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_of(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def pack_weights(struct_cls, fields, params: dict, device):
    """Fill a weights struct from {state_dict key: tensor}; tensors must be fp32, contiguous, on `device`."""
    import torch
    s = struct_cls()
    for f, key in fields:
        t = params[key]
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
            raise SetError("parameter %s must be a contiguous fp32 tensor on %s (got %s, %s, contiguous=%s)"
                           % (key, device, t.dtype, t.device, t.is_contiguous()))
        if t.data_ptr() % 16:
            raise SetError("parameter %s is not 16-byte aligned" % key)
        setattr(s, f, t.data_ptr())
    return s
