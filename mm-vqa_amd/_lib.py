"""ctypes binding of libmmvqa_hip.so (C ABI: include/mmvqa.h).

The product path has NO fallback: if the shared library is missing or fails to
load, every entry point raises.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C mm-vqa_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmvqa_hip.so")

c_f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
c_ptr = C.c_void_p

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SERF, ACT_SILU, ACT_SIGMOID = 0, 1, 2, 3, 4, 5
KIND_FWD, KIND_DGRAD, KIND_WGRAD = 0, 1, 2
PRO_NONE, PRO_AFFINE_RELU, PRO_DZ, PRO_AFFINE, PRO_AFFINE_SILU, PRO_SILU_GATE = 0, 1, 2, 3, 4, 5
EPI_PLAIN, EPI_TAP_FWD, EPI_TAP_BWD = 0, 1, 2
STAT_SLOTS = 16


class BnFold(C.Structure):
    """mirror of mmvqa_bn_fold"""
    _fields_ = [
        ("stat", c_ptr), ("slots", C.c_int), ("bwd", C.c_int), ("publish", C.c_int), ("reps", C.c_int),
        ("count", C.c_double), ("keep", C.c_double), ("eps", C.c_float), ("reserved", C.c_int),
        ("gamma", c_ptr), ("beta", c_ptr), ("mean", c_ptr), ("invstd", c_ptr),
        ("out0", c_ptr), ("out1", c_ptr), ("out2", c_ptr), ("out3", c_ptr),
        ("run_mean", c_ptr), ("run_var", c_ptr), ("nbt", c_ptr), ("dgamma", c_ptr), ("dbeta", c_ptr),
    ]


class GemmDesc(C.Structure):
    """mirror of mmvqa_gemm_desc"""
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("splitk", C.c_int), ("ktiles_per_split", C.c_int),
        ("A", c_ptr), ("A2", c_ptr), ("a_c0", c_ptr), ("a_c1", c_ptr), ("a_c2", c_ptr),
        ("a_pro", C.c_int), ("a_ld", C.c_int),
        ("B", c_ptr), ("b_c0", c_ptr), ("b_c1", c_ptr), ("b_pro", C.c_int), ("b_ld", C.c_int),
        ("b_tapstride", C.c_int),
        ("g_SH", C.c_int), ("g_SW", C.c_int), ("g_Cs", C.c_int), ("g_OH", C.c_int), ("g_OW", C.c_int),
        ("g_KH", C.c_int), ("g_KW", C.c_int), ("g_stride", C.c_int), ("g_pad", C.c_int), ("g_nchw", C.c_int),
        ("C", c_ptr), ("c_ld", C.c_int), ("c_atomic", C.c_int), ("Cpre", c_ptr), ("bias", c_ptr),
        ("act", C.c_int), ("dact", C.c_int), ("Pre", c_ptr), ("pre_ld", C.c_int),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint32), ("R", c_ptr), ("r_ld", C.c_int),
        ("epi_mode", C.c_int), ("tap_HW", C.c_int), ("tap_out", c_ptr), ("tap_dv", c_ptr),
        ("Mk", c_ptr), ("mk_ld", C.c_int), ("mk_s", c_ptr), ("mk_b", c_ptr),
        ("stat1", c_ptr), ("stat_bwd", C.c_int), ("Z1", c_ptr), ("z1_ld", C.c_int), ("mean1", c_ptr),
        ("invstd1", c_ptr),
        ("stat2", c_ptr), ("Z2", c_ptr), ("z2_ld", C.c_int), ("mean2", c_ptr), ("invstd2", c_ptr),
        ("colsum", c_ptr), ("gate", c_ptr), ("gate_hw", C.c_int), ("mk_mode", C.c_int), ("pixmask", c_ptr),
        ("sk_ws", c_ptr), ("sk_ws_floats", C.c_longlong),
        ("a_fold", BnFold), ("stat_slots", C.c_int), ("persist", C.c_int), ("sk_cnt", c_ptr), ("sk_cnt_n", C.c_int),
        ("reserved0", C.c_int),
    ]


class AttnDesc(C.Structure):
    """mirror of mmvqa_attn_desc"""
    _fields_ = [
        ("q", c_ptr), ("k", c_ptr), ("v", c_ptr), ("row_stride", C.c_int), ("head_stride", C.c_int),
        ("out", c_ptr), ("out_row_stride", C.c_int), ("out_head_stride", C.c_int),
        ("mask", c_ptr), ("mask_on_query", C.c_int), ("prev_in", c_ptr), ("prev_out", c_ptr), ("probs", c_ptr),
        ("B", C.c_int), ("T", C.c_int), ("heads", C.c_int), ("sqrt_d", C.c_float), ("drop_p", C.c_float),
        ("seed", C.c_uint32),
        ("dout", c_ptr), ("dq", c_ptr), ("dk", c_ptr), ("dv", c_ptr), ("dprev_in", c_ptr), ("dprev_out", c_ptr),
    ]


class ResampleJob(C.Structure):
    """mirror of mmvqa_resample_job"""
    _fields_ = [
        ("src", c_ptr), ("sh", C.c_int), ("sw", C.c_int), ("spitch", C.c_int),
        ("bx", C.c_int), ("by", C.c_int), ("bw", C.c_int), ("bh", C.c_int),
        ("rw", C.c_int), ("rh", C.c_int), ("ox", C.c_int), ("oy", C.c_int), ("ty0", C.c_int), ("tyn", C.c_int),
        ("tmp", c_ptr), ("dst", c_ptr), ("dpitch", C.c_int),
        ("hb", c_ptr), ("hk", c_ptr), ("hks", C.c_int), ("vb", c_ptr), ("vk", c_ptr), ("vks", C.c_int),
    ]


class ModelDesc(C.Structure):
    """mirror of mmvqa_model_desc"""
    _fields_ = [
        ("cnn", C.c_int), ("resnet_layers", C.c_int * 4), ("resnet_width", C.c_int), ("effnet_depth_div", C.c_int),
        ("encoder", C.c_int), ("hidden", C.c_int), ("heads", C.c_int), ("n_layers", C.c_int),
        ("emb_vocab", C.c_int), ("max_pos", C.c_int), ("type_vocab", C.c_int), ("num_vis", C.c_int),
        ("head_kind", C.c_int), ("n_classes", C.c_int), ("supcon", C.c_int), ("feat_dim", C.c_int),
        ("use_relu", C.c_int), ("p_drop", C.c_float), ("p_emb_drop", C.c_float), ("p_rf_drop", C.c_float),
    ]


# every symbol include/mmvqa.h declares, with (restype, argtypes); checked by tests/test_abi.py
_i, _f, _d, _l, _ll, _u32, _sz = C.c_int, C.c_float, C.c_double, C.c_long, C.c_longlong, C.c_uint32, C.c_size_t
_P = c_ptr
SIGNATURES = {
    "mmvqa_version": (_i, []),
    "mmvqa_last_error": (C.c_char_p, []),
    "mmvqa_sizeof_gemm_desc": (_sz, []),
    "mmvqa_sizeof_attn_desc": (_sz, []),
    "mmvqa_sizeof_model_desc": (_sz, []),
    "mmvqa_igemm": (_i, [C.POINTER(GemmDesc), _i, _i, _i, _P]),
    "mmvqa_attention": (_i, [C.POINTER(AttnDesc), _i, _i, _P]),
    "mmvqa_qkv_attention_fwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _f, _u32]),
    "mmvqa_bn_coef_fwd": (_i, [_P, _P, _i, _d, _f, _P, _P, _P, _P, _P, _f, _i, _i, _P, _P, _P, _P]),
    "mmvqa_bn_coef_bwd": (_i, [_P, _P, _i, _d, _P, _P, _P, _i, _P, _P, _P, _P, _P]),
    "mmvqa_bn_add_relu": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _l, _i]),
    "mmvqa_bn_add_relu_fold": (_i, [_P, _P, C.POINTER(BnFold), _P, C.POINTER(BnFold), _P, _l, _i]),
    "mmvqa_maxpool_fwd": (_i, [_P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _i]),
    "mmvqa_maxpool_bwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _i]),
    "mmvqa_layernorm_fwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _f]),
    "mmvqa_layernorm_bwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i]),
    "mmvqa_embed_fwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _f, _f, _u32]),
    "mmvqa_embed_bwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _f, _u32, _i]),
    "mmvqa_meanpool_fwd": (_i, [_P, _P, _P, _P, _i, _i, _i]),
    "mmvqa_meanpool_bwd": (_i, [_P, _P, _P, _P, _i, _i, _i, _i]),
    "mmvqa_mlm_loss": (_i, [_P, _P, _i, _P, _P, _P, _P, _P, _i, _P, _f, _i, _i, _P]),
    "mmvqa_mlm_grad": (_i, [_P, _P, _i, _P, _P, _P, _i, _P, _f, _i, _i]),
    "mmvqa_asl_loss": (_i, [_P, _P, _i, _P, _P, _P, _i, _i, _i, _f, _f, _f, _f]),
    "mmvqa_l2norm_fwd": (_i, [_P, _P, _P, _P, _i, _i]),
    "mmvqa_l2norm_bwd": (_i, [_P, _P, _P, _P, _P, _i, _i]),
    "mmvqa_supcon_loss": (_i, [_P, _P, _P, _P, _P, _i, _i, _f, _f, _f]),
    "mmvqa_dwconv_fwd": (_i, [_P, _P, _P, _P, _P, _P, _P] + [_i] * 8),
    "mmvqa_dwconv_bwd_data": (_i, [_P] * 14 + [_i] * 8),
    "mmvqa_dwconv_bwd_weight": (_i, [_P] * 10 + [_i] * 8),
    "mmvqa_se_pool": (_i, [_P, _P, _P, _P, _P, _i, _i, _i]),
    "mmvqa_dwconv_fwd_fold": (_i, [_P, _P, _P, _P, _P, _P, _P] + [_i] * 8 + [C.POINTER(BnFold)]),
    "mmvqa_dwconv_bwd_data_fold": (_i, [_P] * 14 + [_i] * 8 + [C.POINTER(BnFold)]),
    "mmvqa_dwconv_bwd_weight_fold": (_i, [_P] * 10 + [_i] * 8 + [C.POINTER(BnFold)]),
    "mmvqa_se_pool_fold": (_i, [_P, _P, _P, _P, _P, _i, _i, _i, C.POINTER(BnFold)]),
    "mmvqa_bn_act_add_fold": (_i, [_P, _P, C.POINTER(BnFold), _i, _P, C.POINTER(BnFold), _i, _P, _l, _i]),
    "mmvqa_se_dgate": (_i, [_P, _P, _P, _P, _P, _P, _i, _i, _i]),
    "mmvqa_tap_thin_ok": (_i, [_l, _i, _i, _i]),
    "mmvqa_tap_thin_fwd": (_i, [_P] * 6 + [_l, _i, _i, _i, _i]),
    "mmvqa_tap_thin_bwd": (_i, [_P] * 7 + [_l, _i, _i, _i, _i]),
    "mmvqa_se_fc_fwd": (_i, [_P] * 10 + [_i, _i, _i]),
    "mmvqa_se_fc_bwd_scratch_floats": (_sz, [_i, _i, _i]),
    "mmvqa_se_fc_bwd": (_i, [_P] * 14 + [_i, _i, _i]),
    "mmvqa_act_bwd_stats": (_i, [_P] * 9 + [_i, _P, _P, _l, _i, _i]),
    "mmvqa_bn_act_add": (_i, [_P, _P, _P, _P, _i, _P, _P, _P, _i, _i, _P, _l, _i]),
    "mmvqa_adam": (_i, [_P, _P, _P, _P, _P, _l, _d, _d, _d, _d, _i, _f, _i]),
    "mmvqa_axpy": (_i, [_P, _P, _P, _f, _l]),
    "mmvqa_colsum": (_i, [_P, _P, _i, _i, _i, _P]),
    "mmvqa_dropout": (_i, [_P, _P, _l, _f, _u32]),
    "mmvqa_pixmask": (_i, [_P, _P] + [_i] * 9),
    "mmvqa_sizeof_resample_job": (_sz, []),
    "mmvqa_resample_coeffs": (_i, [_i, _d, _d, _i, _P, _P, _i]),
    "mmvqa_aug_resample": (_i, [_P, _P, _i, _i, _i, _i]),
    "mmvqa_aug_rotate": (_i, [_P, _P, _P, _P, _i, _i, _i]),
    "mmvqa_aug_jitter_round": (_i, [_P, _P, _P, _P, _P, _i, _i]),
    "mmvqa_aug_to_tensor": (_i, [_P, _P, _P, _i, _i, _P, _P]),
    "mmvqa_engine_create": (_i, [C.POINTER(ModelDesc), C.POINTER(_P)]),
    "mmvqa_engine_destroy": (None, [_P]),
    "mmvqa_engine_num_tensors": (_i, [_P]),
    "mmvqa_engine_tensor_info": (_i, [_P, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_ll * 4),
                                      C.POINTER(_ll), C.POINTER(_i)]),
    "mmvqa_engine_param_floats": (_ll, [_P]),
    "mmvqa_engine_buf_floats": (_ll, [_P]),
    "mmvqa_engine_nbt_count": (_ll, [_P]),
    "mmvqa_engine_plan": (_sz, [_P, _i, _i, _i, _i]),
    "mmvqa_engine_bind": (_i, [_P, _P, _P, _P, _P, _P, _sz]),
    "mmvqa_engine_forward": (_i, [_P, _P, _P, _P, _P, _P, _P, _i, _P, _i, _u32]),
    "mmvqa_engine_backward": (_i, [_P, _P, _P, _i, _P]),
    "mmvqa_engine_set_grad_callback": (_i, [_P, _P, _P]),
    "mmvqa_engine_tune": (_i, [_P, _i]),
    "mmvqa_engine_profile": (_i, [_P, _i]),
    "mmvqa_engine_profile_read": (_i, [_P, _i, C.POINTER(_ll), C.POINTER(_d), C.POINTER(_d)]),
    "mmvqa_engine_profile_read_region": (_i, [_P, _i, _i, C.POINTER(_ll), C.POINTER(_d), C.POINTER(_d)]),
    "mmvqa_engine_profile_read_hbm": (_i, [_P, _i, C.POINTER(_ll), C.POINTER(_d), C.POINTER(_d)]),
}

_lib = None


class MMVQAError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MMVQAError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback -- run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, no GPU required).")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    for nm, st in (("gemm", GemmDesc), ("attn", AttnDesc), ("model", ModelDesc), ("resample_job", ResampleJob)):
        if nm == "resample_job":
            got = L.mmvqa_sizeof_resample_job()
            if got != C.sizeof(st):
                raise MMVQAError(f"ABI mismatch: sizeof(mmvqa_resample_job) = {got} in the library, {C.sizeof(st)} in Python")
            continue
        got = getattr(L, f"mmvqa_sizeof_{nm}_desc")()
        if got != C.sizeof(st):
            raise MMVQAError(f"ABI mismatch: sizeof(mmvqa_{nm}_desc) = {got} in the library, {C.sizeof(st)} in Python")
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        msg = lib().mmvqa_last_error()
        raise MMVQAError(f"mmvqa error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """device/host pointer of a tensor (None -> NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
